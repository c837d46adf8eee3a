// jmhip_stubs.hip -- entry points declared in include/jmhip.h whose device stage is not built yet:
// they fail loudly (never a CPU fallback).
#include "jmhip_internal.h"

extern "C" int jmhip_tq_batch(jmhip_ctx *c, int, int, const jmhip_quant *, int, const jmhip_tq_job *, int, jmhip_tq_result *)
{ return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_tq_batch: not built yet"); }
extern "C" void jmhip_flat_quant(jmhip_quant *, int, int, int) {}
extern "C" int jmhip_distortion_batch(jmhip_ctx *c, const jmhip_dist_job *, int, int32_t *)
{ return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_distortion_batch: not built yet"); }
