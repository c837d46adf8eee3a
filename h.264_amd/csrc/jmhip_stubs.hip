// jmhip_stubs.hip -- entry points declared in include/jmhip.h whose device stage is not built yet:
// they fail loudly (never a CPU fallback).
#include "jmhip_internal.h"

extern "C" int jmhip_tq_batch(jmhip_ctx *c, int, int, const jmhip_quant *, int, const jmhip_tq_job *, int, jmhip_tq_result *)
{ return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_tq_batch: not built yet"); }
extern "C" void jmhip_flat_quant(jmhip_quant *, int, int, int) {}
extern "C" int jmhip_me_frame(jmhip_ctx *c, const jmhip_me_params *, const jmhip_me_mb *, int, jmhip_me_result *)
{ return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_me_frame: not built yet"); }
extern "C" int jmhip_me_frame_async(jmhip_ctx *c, const jmhip_me_params *, const jmhip_me_mb *, int)
{ return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_me_frame_async: not built yet"); }
extern "C" int jmhip_me_results_download(jmhip_ctx *c, jmhip_me_result *, int)
{ return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_me_results_download: not built yet"); }
extern "C" int jmhip_distortion_batch(jmhip_ctx *c, const jmhip_dist_job *, int, int32_t *)
{ return jm_fail(c, JMHIP_ERR_UNSUPPORTED, "jmhip_distortion_batch: not built yet"); }
extern "C" void jmhip_partition_info(int, int *, int *, int *, int *, int *) {}
