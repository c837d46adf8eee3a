// jmhip_stubs.hip -- entry points declared in include/jmhip.h whose device stage is not built yet:
// they fail loudly (never a CPU fallback).
#include "jmhip_internal.h"

extern "C" int jmhip_distortion_batch(jmhip_ctx *c, const jmhip_dist_job *, int, int32_t *)
{ return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_distortion_batch: not built yet"); }
