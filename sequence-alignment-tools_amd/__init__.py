"""MI355X-native multi-pattern DNA matcher: Python host side over the C ABI (include/pm_gpu.h).

The package directory name contains '-', so import it through the repo-root shim:
    import sat_amd
    pm = sat_amd.PatternMatch(k=2, indels=False)

`PatternMatch` mirrors the reference's operator surface (reference pattern_match.h:84-156):
add_pattern / init / find_patterns / reset, same argument meaning, same hit triples.  The
library is the in-tree HIP build (csrc/libpm_gpu.so); there is no CPU fallback -- if the library
or a gfx950 device is missing, calls fail loudly.
"""
from .pattern_match import (  # noqa: F401
    HIT_DTYPE, PM_E_OVERFLOW, KERNEL_AUTO, KERNEL_BITPAR, KERNEL_SEED, SEM_AUTO, SEM_EXACT_BASES, SEM_EXACT_HALVES,
    SEM_FILTER_BITVEC, SEM_KEYWORD_TREE, SEM_SHIFT_AND, SEM_SHIFT_AND_INEXACT, PatternMatch, PmError,
    build_library, library_path, load_library, measure_stream_read, pick_semantics, reverse_comp, sorted_tuples,
)
