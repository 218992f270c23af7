// env_knobs.hpp -- environment variables the library reads (the one table of them is DESIGN.md section 10).
// Three kinds: (1) settings a deployment may use and (2) hooks the tests / tools need (forcing a path or a size on small inputs, phase
// timings on stderr) are read with getenv in every build; (3) A/B EXPERIMENT knobs -- alternative kernels and thresholds kept for
// measurements -- are read through cmi_exp_env and exist only in builds made with `make EXP=1` (-DCMI_EXPERIMENT_KNOBS): the release
// library ignores them.  No knob changes a result (the one that did, CMI_DEBUG_MERGE_LEVELS, additionally needs -DCMI_TIMING_EXPERIMENTS).
#pragma once
#include <cstdlib>

inline const char *cmi_exp_env(const char *name) {
#ifdef CMI_EXPERIMENT_KNOBS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
