#!/bin/bash
# Everything a round's profiles/ set is made of, in one call on the GPU box (about 12 minutes):
#   bench lines per option set (bench_lines.sh), rocprofv3 kernel statistics (profile_round.sh's part), fabric traffic and
#   pipe counters of the two headline kernels (pmc_traffic.sh, pmc_issue.sh: separate --pmc passes, nothing else on the
#   command line), the command lines at 3 Gbp (cli_scale.py), pm_scan in ranges of three sizes, the edit plan's A/B.
# Usage (GPU box): bash scripts/round_profiles.sh r04      then, in the build container: python scripts/collect_profiles.py r04
tag=${1:-r04}
part=${2:-all}           # "lines": bench lines, kernel traces, pm_scan ranges, command lines; "pmc": the counter passes; "all": both (over 20 minutes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$tag gpurun_out/lines_$tag
if [ "$part" = pmc ] || [ "$part" = all ]; then
  # fabric traffic and pipe counters of every option set's scan kernel: nothing in the bench lines may point at counters of other code
  for spec in "K2 pm_pair_scan --k 2" "k2_edits pm_pair_edit_scan --k 2 --indels 1" "K0 pm_seed_scan --k 0" "K1 pm_pair_scan --k 1" "k1_edits pm_half_scan --k 1 --indels 1"; do
    set -- $spec; name=$1; kern=$2; shift 2
    bash scripts/pmc_traffic.sh $tag $name $kern "$@" | tail -1
    bash scripts/pmc_issue.sh $tag $name $kern "$@" | tail -1
  done
  [ "$part" = pmc ] && { echo "round profiles (pmc) done"; exit 0; }
fi
bash scripts/bench_lines.sh $tag 2>&1 | tail -12
for spec in "K2 --k 2" "K0 --k 0" "K1 --k 1" "K2_1M --primers 1000000"; do
  set -- $spec; name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag/kt_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu --scan-passes 0 "$@" > gpurun_out/prof_$tag/kt_$name.log 2>&1
  echo "kernel-trace $name rc=$?"
done
for c in 26 28 30; do
  python3 bench.py --no-cpu --steps 5 --warmup 2 --scan-chunk $((1<<c)) > gpurun_out/lines_$tag/bench_K2_scanchunk$c.json 2>/dev/null
done
python3 scripts/edit_pair_floor.py > gpurun_out/${tag}_edit_pair_floor.json 2> gpurun_out/${tag}_edit_pair_floor.err
python3 scripts/cli_scale.py --bases 3000000000 --primers 100000 --pairs 100000 > gpurun_out/${tag}_cli_scale_3g.json 2> gpurun_out/${tag}_cli_scale_3g.err
echo "cli rc=$?"
echo "round profiles done"
