run() { env "$@" timeout -k 10 200 python bench.py --db-bases 1000000000 --steps 3 --warmup 1 --k 2 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'kernel ms', round(d['roofline']['kernel_ms'],2))"; }
run X=1
run PM_SEED_GROUP=100000
run PM_SEED_GROUP=64
run PM_SEED_GROUP=1024
run PM_SEED_CHUNK=262144
run PM_SEED_CHUNK=1048576
run PM_SEED_CHUNK=2097152 PM_SEED_GROUP=100000
run PM_SEED_DEBUG=1
