// switches.h — the ONE place libswmarlin.so reads its environment ("geometry knobs nobody tests are a way to get
// wrong commitments from a typo").
//
// Until r05 some seventy SWM_* variables were read where they were used, most of them the launch geometry of an experiment.  What
// is left falls in two classes, and tests/test_switch_audit.py holds the source to that:
//   * switches that select between MAINTAINED paths of the product — the twisted Edwards / XYZZ table forms, the precomputed-window
//     / per-window schedules, the table width, the low-LDS and quad bucket stages, the one-stream schedule of small proofs, the
//     mask commitment in pieces, the lazy transform and its per-pass tables, the sharding forms — every one exercised for golden
//     proof bytes by tests/test_gpu_switches.py;
//   * diagnostics that cannot change a result (SWM_TRACE, SWM_PROOF_MARKS, the host pool's size and spin, SWM_RCCL_PATH).
// Every value is an integer with a declared range: a value that does not parse, or lies outside it, is REFUSED — one line on
// stderr and the default is used — it is never clamped into a geometry nobody asked for.
// The measurement hook of tools/ubench/shard_emulate.py (exchanges answered with the rank's own data: wrong proofs by
// construction) is compiled only with -DSWM_MEASURE_HOOKS, into a second library that tool builds; the shipped library has no
// switch that can make swm_generate_proof return a wrong proof with status 0.
#pragma once
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>

namespace swm {

// SWM_<...> as an integer in [lo, hi]; unset or empty: def.
inline long env_switch(const char* name, long def, long lo, long hi) {
    const char* e = getenv(name);
    if (!e || !*e) return def;
    char* end = nullptr;
    errno = 0;
    const long v = strtol(e, &end, 0);
    if (errno != 0 || end == e || *end != '\0' || v < lo || v > hi) {
        fprintf(stderr, "[swm] %s=%s refused (an integer in [%ld, %ld] is expected): using the default %ld\n", name, e, lo, hi, def);
        return def;
    }
    return v;
}
// set to anything but "" / "0"
inline bool env_flag(const char* name) {
    const char* e = getenv(name);
    return e && *e && !(e[0] == '0' && e[1] == '\0');
}
// a path (SWM_RCCL_PATH); nullptr when unset or empty
inline const char* env_path(const char* name) {
    const char* e = getenv(name);
    return e && *e ? e : nullptr;
}

}  // namespace swm
