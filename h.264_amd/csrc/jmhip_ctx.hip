// jmhip_ctx.hip -- context, device pictures, upload/download, stage timing (C ABI entry points)
#include "jmhip_internal.h"
#include <atomic>
#include <thread>

static void chroma_geometry(int yuv, ChromaGeom *g)   // lencod/src/lencod.c:2851-2884, img_chroma.c:390-405
{
  memset(g, 0, sizeof(*g));
  if (yuv == JMHIP_YUV420)      *g = {8, 8, 10, 10, 3, 3, 7, 7, 1, 1, 8, 8};
  else if (yuv == JMHIP_YUV422) *g = {8, 4, 10, 20, 3, 2, 7, 3, 1, 2, 8, 16};
  else if (yuv == JMHIP_YUV444) *g = {4, 4, 20, 20, 2, 2, 3, 3, 2, 2, 16, 16};
}

extern "C" int jmhip_abi_version(void) { return JMHIP_ABI_VERSION; }

extern "C" int jmhip_sizeof(int which)
{
  switch (which) {
  case 0: return (int)sizeof(jmhip_me_mb);
  case 1: return (int)sizeof(jmhip_me_result);
  case 2: return (int)sizeof(jmhip_quant);
  case 3: return (int)sizeof(jmhip_tq_job);
  case 4: return (int)sizeof(jmhip_tq_result);
  case 5: return (int)sizeof(jmhip_dist_job);
  case 6: return (int)sizeof(jmhip_me_params);
  case 7: return (int)sizeof(jmhip_config);
  case 8: return (int)sizeof(jmhip_mb_mode);
  case 9: return (int)sizeof(jmhip_surface_job);
  case 10: return (int)sizeof(jmhip_bipred_job);
  case 11: return (int)sizeof(jmhip_bipred_result);
  case 12: return (int)sizeof(jmhip_bipred_params);
  case 13: return (int)sizeof(jmhip_predcost_job);
  case 14: return (int)sizeof(jmhip_deblock_mb);
  case 15: return (int)sizeof(jmhip_deblock_blk);
  case 16: return (int)sizeof(jmhip_deblock_params);
  case 17: return (int)sizeof(jmhip_slice_params);
  case 18: return (int)sizeof(jmhip_mb_inter);
  case 19: return (int)sizeof(jmhip_frame_wp);
  case 20: return (int)sizeof(jmhip_mb_bipred);
  case 21: return (int)sizeof(jmhip_frame_bw);
  case 22: return (int)sizeof(jmhip_mb_residual);
  default: return -1;
  }
}

extern "C" const char *jmhip_strerror(int code)
{
  switch (code) {
  case JMHIP_OK: return "ok";
  case JMHIP_ERR_ARG: return "bad argument";
  case JMHIP_ERR_DEVICE: return "HIP device error";
  case JMHIP_ERR_UNSUPPORTED: return "configuration not supported by this version";
  case JMHIP_ERR_NOMEM: return "out of device memory";
  default: return "unknown error";
  }
}

extern "C" const char *jmhip_last_error(jmhip_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

extern "C" int jmhip_ctx_create(const jmhip_config *cfg, jmhip_ctx **out)
{
  if (!cfg || !out) return JMHIP_ERR_ARG;
  *out = nullptr;
  if (cfg->width <= 0 || cfg->height <= 0 || (cfg->width & 15) || (cfg->height & 15)) return JMHIP_ERR_ARG;
  if (cfg->yuv_format < JMHIP_YUV400 || cfg->yuv_format > JMHIP_YUV444) return JMHIP_ERR_ARG;
  if (cfg->bit_depth != 8) return JMHIP_ERR_UNSUPPORTED;
  if (cfg->max_refs < 1 || cfg->max_refs > 32 || cfg->search_range < 0 || cfg->search_range > 64) return JMHIP_ERR_ARG;

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return JMHIP_ERR_DEVICE;
  if (cfg->device < 0 || cfg->device >= ndev) return JMHIP_ERR_ARG;
  if (hipSetDevice(cfg->device) != hipSuccess) return JMHIP_ERR_DEVICE;

  jmhip_ctx *c = new jmhip_ctx();
  c->cfg = *cfg;
  c->W = cfg->width; c->H = cfg->height;
  c->Wp = c->W + 2 * JMHIP_PAD; c->Hp = c->H + 2 * JMHIP_PAD;
  c->mbw = c->W / 16; c->mbh = c->H / 16;
  chroma_geometry(cfg->yuv_format, &c->cg);
  if (cfg->yuv_format != JMHIP_YUV400) {
    c->Wc = (cfg->yuv_format == JMHIP_YUV444) ? c->W : c->W / 2;
    c->Hc = (cfg->yuv_format == JMHIP_YUV420) ? c->H / 2 : c->H;
    c->Wcp = c->Wc + 2 * c->cg.pad_x; c->Hcp = c->Hc + 2 * c->cg.pad_y;
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return JMHIP_ERR_DEVICE; }
  c->refs.resize(cfg->max_refs);

  auto alloc = [&](uint8_t **p, size_t bytes) -> bool {
    if (hipMalloc((void **)p, bytes) != hipSuccess) return false;
    return hipMemsetAsync(*p, 0, bytes, c->stream) == hipSuccess;
  };
  bool ok = true;
  const size_t ysz = (size_t)c->W * c->H, csz = (size_t)c->Wc * c->Hc;
  for (auto &r : c->refs) {
    ok = ok && alloc(&r.y, ysz);
    ok = ok && alloc(&r.luma_sub, 16 * (size_t)c->Wp * c->Hp + 64);   // +64: unaligned row fetches read up to 11 bytes past a row start
    if (csz) {
      ok = ok && alloc(&r.u, csz) && alloc(&r.v, csz);
      // zero-initialised like JM's calloc (memalloc.c:142): the last row/column are never written
      for (int k = 0; k < 2; k++) ok = ok && alloc(&r.cr_sub[k], (size_t)c->cg.sub_x * c->cg.sub_y * c->Wcp * c->Hcp);
    }
  }
  ok = ok && alloc(&c->cur_own[0], ysz);
  if (csz) ok = ok && alloc(&c->cur_own[1], csz) && alloc(&c->cur_own[2], csz);
  c->cur_y = c->cur_own[0]; c->cur_u = c->cur_own[1]; c->cur_v = c->cur_own[2];
  if (!ok) { jmhip_ctx_destroy(c); return JMHIP_ERR_NOMEM; }
  if (hipStreamSynchronize(c->stream) != hipSuccess) { jmhip_ctx_destroy(c); return JMHIP_ERR_DEVICE; }
  *out = c;
  return JMHIP_OK;
}

extern "C" void jmhip_ctx_destroy(jmhip_ctx *c)
{
  if (!c) return;
  (void)hipSetDevice(c->cfg.device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto &r : c->refs) {
    (void)hipFree(r.y); (void)hipFree(r.u); (void)hipFree(r.v); (void)hipFree(r.luma_sub);
    (void)hipFree(r.cr_sub[0]); (void)hipFree(r.cr_sub[1]);
  }
  (void)hipFree(c->cur_own[0]); (void)hipFree(c->cur_own[1]); (void)hipFree(c->cur_own[2]);
  if (c->pin_host) (void)hipHostFree(c->pin_host);
  for (auto e : c->pin_evt) (void)hipEventDestroy(e);
  (void)hipFree(c->stage_dev); (void)hipFree(c->me_jobs_dev); (void)hipFree(c->me_res_dev); (void)hipFree(c->ref_ptrs_dev); (void)hipFree(c->me_idx_dev); (void)hipFree(c->surf_dev); (void)hipFree(c->surf_jobs_dev);
  (void)hipFree(c->tq_jobs_dev); (void)hipFree(c->tq_res_dev); (void)hipFree(c->tq_quant_dev);
  (void)hipFree(c->fr_bi); (void)hipFree(c->fr_rec); (void)hipFree(c->fr_blk_ref); (void)hipFree(c->fr_jobs_y); (void)hipFree(c->fr_jobs_c); (void)hipFree(c->fr_res_y); (void)hipFree(c->fr_res_c);
  jm_slice_state_free(c);
  jm_xslice_free(c);
  (void)hipFree(c->dbk_dev); (void)hipFree(c->dbr_dev); (void)hipFree(c->fr_quant); (void)hipFree(c->fr_modes); (void)hipFree(c->rec_y); (void)hipFree(c->pred_y); (void)hipFree(c->pred_u); (void)hipFree(c->pred_v); (void)hipFree(c->rec_u); (void)hipFree(c->rec_v);
  for (auto &p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto e : c->evt_pool) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// ---------------------------------------------------------------------------------------- timing

static int drain_timing(jmhip_ctx *c)
{
  for (auto &p : c->pending) {
    float ms = 0.f;
    JM_HIP_CHECK(c, hipEventSynchronize(p.b));
    JM_HIP_CHECK(c, hipEventElapsedTime(&ms, p.a, p.b));
    c->stage_ms[p.stage] += ms;
    c->stage_launches[p.stage] += 1;
    c->evt_pool.push_back(p.a); c->evt_pool.push_back(p.b);
  }
  c->pending.clear();
  return JMHIP_OK;
}

static hipEvent_t get_evt(jmhip_ctx *c)
{
  if (!c->evt_pool.empty()) { hipEvent_t e = c->evt_pool.back(); c->evt_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void jm_stage_begin(jmhip_ctx *c, int stage)
{
  if (!c->timing || !((c->timing_mask >> stage) & 1)) return;
  jmhip_ctx::PendingEvt p{stage, get_evt(c), get_evt(c)};
  (void)hipEventRecord(p.a, c->stream);
  c->pending.push_back(p);
}

void jm_stage_end(jmhip_ctx *c, int stage)
{
  if (!c->timing || !((c->timing_mask >> stage) & 1) || c->pending.empty()) return;
  (void)hipEventRecord(c->pending.back().b, c->stream);
}

extern "C" int jmhip_sync(jmhip_ctx *c)
{
  if (!c) return JMHIP_ERR_ARG;
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return JMHIP_OK;
}

extern "C" void *jmhip_stream_handle(jmhip_ctx *c) { return c ? (void *)c->stream : nullptr; }

extern "C" int jmhip_timing_enable(jmhip_ctx *c, int on)
{
  if (!c) return JMHIP_ERR_ARG;
  c->timing = on != 0;
  return JMHIP_OK;
}

extern "C" int jmhip_timing_select(jmhip_ctx *c, unsigned stage_mask)
{
  if (!c) return JMHIP_ERR_ARG;
  c->timing_mask = stage_mask;
  return JMHIP_OK;
}

extern "C" int jmhip_timing_read(jmhip_ctx *c, double ms[JMHIP_STAGE_COUNT], int launches[JMHIP_STAGE_COUNT])
{
  if (!c || !ms || !launches) return JMHIP_ERR_ARG;
  int rc = drain_timing(c);
  if (rc) return rc;
  for (int i = 0; i < JMHIP_STAGE_COUNT; i++) {
    ms[i] = c->stage_ms[i]; launches[i] = c->stage_launches[i];
    c->stage_ms[i] = 0; c->stage_launches[i] = 0;
  }
  return JMHIP_OK;
}

// ---------------------------------------------------------------------------------------- pictures

__global__ void narrow_u16_kernel(const uint16_t *__restrict__ src, uint8_t *__restrict__ dst, int w, int h, int sstride)
{
  int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x < w && y < h) dst[(size_t)y * w + x] = (uint8_t)src[(size_t)y * sstride + x];
}

__global__ void widen_u8_kernel(const uint8_t *__restrict__ src, uint16_t *__restrict__ dst, size_t n)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

static int ensure_stage(jmhip_ctx *c, size_t bytes)
{
  if (c->stage_bytes >= bytes) return JMHIP_OK;
  if (c->stage_dev) JM_HIP_CHECK(c, hipFree(c->stage_dev));
  c->stage_dev = nullptr; c->stage_bytes = 0;
  if (hipMalloc(&c->stage_dev, bytes) != hipSuccess) return jm_fail(c, JMHIP_ERR_NOMEM, "staging buffer");
  c->stage_bytes = bytes;
  return JMHIP_OK;
}

// copy one plane (w x h samples) into a tightly packed 8-bit device plane
// the recon picture: all three planes or none (a failed chroma allocation must not leave a luma plane that later calls take
// as "allocated" and launch on null chroma pointers)
int jm_ensure_recon(jmhip_ctx *c)
{
  if (c->rec_y && (!c->Wc || (c->rec_u && c->rec_v))) return JMHIP_OK;
  bool ok = hipMalloc((void **)&c->rec_y, (size_t)c->W * c->H) == hipSuccess;
  if (ok && c->Wc) ok = hipMalloc((void **)&c->rec_u, (size_t)c->Wc * c->Hc) == hipSuccess && hipMalloc((void **)&c->rec_v, (size_t)c->Wc * c->Hc) == hipSuccess;
  if (!ok) {
    if (c->rec_y) (void)hipFree(c->rec_y);
    if (c->rec_u) (void)hipFree(c->rec_u);
    if (c->rec_v) (void)hipFree(c->rec_v);
    c->rec_y = c->rec_u = c->rec_v = nullptr;
    return jm_fail(c, JMHIP_ERR_NOMEM, "recon picture");
  }
  return JMHIP_OK;
}

int jm_upload_plane(jmhip_ctx *c, uint8_t *dst, const void *src, int w, int h, int pel_bytes, int stride, int device_ptrs)
{
  if (!src) return jm_fail(c, JMHIP_ERR_ARG, "NULL plane");
  if (stride < w) return jm_fail(c, JMHIP_ERR_ARG, "stride < width");
  if (pel_bytes == 1) {
    JM_HIP_CHECK(c, hipMemcpy2DAsync(dst, w, src, stride, w, h, device_ptrs ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    // host buffers are only borrowed for the duration of the call (SURVEY 8(b) ownership): the copy must have left them
    if (!device_ptrs) JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    return JMHIP_OK;
  }
  if (pel_bytes != 2 || device_ptrs) return jm_fail(c, JMHIP_ERR_ARG, "pel_bytes must be 1 (or 2 for host pointers)");
  size_t bytes = (size_t)stride * h * 2;
  int rc = ensure_stage(c, bytes);
  if (rc) return rc;
  JM_HIP_CHECK(c, hipMemcpyAsync(c->stage_dev, src, bytes, hipMemcpyHostToDevice, c->stream));
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));    // the caller's buffer is only borrowed for the call
  dim3 grid((w + 255) / 256, h);
  narrow_u16_kernel<<<grid, 256, 0, c->stream>>>((const uint16_t *)c->stage_dev, dst, w, h, stride);
  JM_HIP_CHECK(c, hipGetLastError());
  // the staging buffer is reused by the next plane: order is kept by the stream
  return JMHIP_OK;
}

extern "C" int jmhip_ref_upload(jmhip_ctx *c, int ref, const void *Y, const void *U, const void *V,
                                int pel_bytes, int stride_y, int stride_c, int device_ptrs)
{
  if (!c) return JMHIP_ERR_ARG;
  if (ref < 0 || ref >= (int)c->refs.size()) return jm_fail(c, JMHIP_ERR_ARG, "ref slot out of range");
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  RefSlot &r = c->refs[ref];
  int rc = jm_upload_plane(c, r.y, Y, c->W, c->H, pel_bytes, stride_y, device_ptrs);
  if (rc) return rc;
  if (c->Wc) {
    if ((rc = jm_upload_plane(c, r.u, U, c->Wc, c->Hc, pel_bytes, stride_c, device_ptrs))) return rc;
    if ((rc = jm_upload_plane(c, r.v, V, c->Wc, c->Hc, pel_bytes, stride_c, device_ptrs))) return rc;
  }
  r.has_pic = true; r.has_luma_sub = false; r.has_cr_sub = false;
  return JMHIP_OK;
}

extern "C" int jmhip_cur_upload(jmhip_ctx *c, const void *Y, const void *U, const void *V,
                                int pel_bytes, int stride_y, int stride_c, int device_ptrs)
{
  if (!c) return JMHIP_ERR_ARG;
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  c->cur_y = c->cur_own[0]; c->cur_u = c->cur_own[1]; c->cur_v = c->cur_own[2];
  int rc = jm_upload_plane(c, c->cur_y, Y, c->W, c->H, pel_bytes, stride_y, device_ptrs);
  if (rc) return rc;
  if (c->Wc && U && V) {
    if ((rc = jm_upload_plane(c, c->cur_u, U, c->Wc, c->Hc, pel_bytes, stride_c, device_ptrs))) return rc;
    if ((rc = jm_upload_plane(c, c->cur_v, V, c->Wc, c->Hc, pel_bytes, stride_c, device_ptrs))) return rc;
  }
  c->has_cur = true;
  return JMHIP_OK;
}

extern "C" int jmhip_ref_planes_peek(jmhip_ctx *c, int ref, void **Y, void **U, void **V, int *pitch_y, int *pitch_c)
{
  if (!c) return JMHIP_ERR_ARG;
  if (ref < 0 || ref >= (int)c->refs.size()) return jm_fail(c, JMHIP_ERR_ARG, "ref slot out of range");
  const RefSlot &r = c->refs[ref];
  if (Y) *Y = r.y;
  if (U) *U = r.u;
  if (V) *V = r.v;
  if (pitch_y) *pitch_y = c->W;
  if (pitch_c) *pitch_c = c->Wc;
  return JMHIP_OK;
}

extern "C" int jmhip_copy_from_device(jmhip_ctx *c, const void *device_src, void *host_dst, size_t bytes)
{
  if (!c || !device_src || !host_dst) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_copy_from_device: NULL") : JMHIP_ERR_ARG;
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  JM_HIP_CHECK(c, hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, c->stream));
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return JMHIP_OK;
}

extern "C" int jmhip_cur_bind(jmhip_ctx *c, const void *Y, const void *U, const void *V)
{
  if (!c || !Y || (c->Wc && (!U || !V))) return c ? jm_fail(c, JMHIP_ERR_ARG, "jmhip_cur_bind: NULL plane") : JMHIP_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(V)) & 3) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_cur_bind: planes must be 4-byte aligned");
  c->cur_y = (uint8_t *)Y; c->cur_u = (uint8_t *)U; c->cur_v = (uint8_t *)V;
  c->has_cur = true;
  return JMHIP_OK;
}

extern "C" int jmhip_ref_device_planes(jmhip_ctx *c, int ref, void **Y, void **U, void **V, int *pitch_y, int *pitch_c)
{
  if (!c) return JMHIP_ERR_ARG;
  if (ref < 0 || ref >= (int)c->refs.size()) return jm_fail(c, JMHIP_ERR_ARG, "ref slot out of range");
  RefSlot &r = c->refs[ref];
  if (Y) *Y = r.y;
  if (U) *U = r.u;
  if (V) *V = r.v;
  if (pitch_y) *pitch_y = c->W;
  if (pitch_c) *pitch_c = c->Wc;
  r.has_pic = true;          // the caller writes the planes itself (device-to-device exchange)
  r.has_luma_sub = false; r.has_cr_sub = false;
  return JMHIP_OK;
}

int jm_download_planes(jmhip_ctx *c, const uint8_t *src, size_t n, void *out, int pel_bytes)
{
  if (!out) return jm_fail(c, JMHIP_ERR_ARG, "NULL output");
  if (pel_bytes == 1) {
    JM_HIP_CHECK(c, hipMemcpyAsync(out, src, n, hipMemcpyDeviceToHost, c->stream));
  } else if (pel_bytes == 2) {
    int rc = ensure_stage(c, n * 2);
    if (rc) return rc;
    widen_u8_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(src, (uint16_t *)c->stage_dev, n);
    JM_HIP_CHECK(c, hipGetLastError());
    JM_HIP_CHECK(c, hipMemcpyAsync(out, c->stage_dev, n * 2, hipMemcpyDeviceToHost, c->stream));
  } else return jm_fail(c, JMHIP_ERR_ARG, "pel_bytes must be 1 or 2");
  JM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return JMHIP_OK;
}

extern "C" int jmhip_ref_download_luma(jmhip_ctx *c, int ref, void *out, int pel_bytes)
{
  if (!c) return JMHIP_ERR_ARG;
  if (ref < 0 || ref >= (int)c->refs.size()) return jm_fail(c, JMHIP_ERR_ARG, "ref slot out of range");
  if (!c->refs[ref].has_luma_sub) return jm_fail(c, JMHIP_ERR_ARG, "luma sub-pel planes not built (call jmhip_interp_luma)");
  return jm_download_planes(c, c->refs[ref].luma_sub, 16 * (size_t)c->Wp * c->Hp, out, pel_bytes);
}

extern "C" int jmhip_ref_download_chroma(jmhip_ctx *c, int ref, int uv, void *out, int pel_bytes)
{
  if (!c) return JMHIP_ERR_ARG;
  if (ref < 0 || ref >= (int)c->refs.size() || uv < 0 || uv > 1) return jm_fail(c, JMHIP_ERR_ARG, "ref/uv out of range");
  if (!c->Wc) return jm_fail(c, JMHIP_ERR_ARG, "4:0:0 has no chroma");
  if (!c->refs[ref].has_cr_sub) return jm_fail(c, JMHIP_ERR_ARG, "chroma sub-pel planes not built (call jmhip_interp_chroma)");
  return jm_download_planes(c, c->refs[ref].cr_sub[uv], (size_t)c->cg.sub_x * c->cg.sub_y * c->Wcp * c->Hcp, out, pel_bytes);
}

// Planes to the caller's own row pointers (JM keeps every plane as an array of rows, imgpel **): each plane travels as packed bytes into a
// page-locked staging buffer, and while the next planes are still on the link the host widens / copies the arrived one row by row.
static int download_planes_rows(jmhip_ctx *c, const uint8_t *src, int planes, int rows, int width, void *const *out_rows, int pel_bytes)
{
  if (!out_rows) return jm_fail(c, JMHIP_ERR_ARG, "NULL row table");
  if (pel_bytes != 1 && pel_bytes != 2) return jm_fail(c, JMHIP_ERR_ARG, "pel_bytes must be 1 or 2");
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  const size_t plane = (size_t)rows * width, n = plane * planes;
  if (c->pin_bytes < n) {
    if (c->pin_host) JM_HIP_CHECK(c, hipHostFree(c->pin_host));
    c->pin_host = nullptr; c->pin_bytes = 0;
    if (hipHostMalloc((void **)&c->pin_host, n, hipHostMallocDefault) != hipSuccess) return jm_fail(c, JMHIP_ERR_NOMEM, "page-locked staging buffer");
    c->pin_bytes = n;
  }
  while ((int)c->pin_evt.size() < planes) {
    hipEvent_t e;
    JM_HIP_CHECK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->pin_evt.push_back(e);
  }
  for (int p = 0; p < planes; p++) {
    if (!out_rows[(size_t)p * rows]) continue;                       // a plane whose first row pointer is NULL is not wanted
    JM_HIP_CHECK(c, hipMemcpyAsync(c->pin_host + p * plane, src + p * plane, plane, hipMemcpyDeviceToHost, c->stream));
    JM_HIP_CHECK(c, hipEventRecord(c->pin_evt[p], c->stream));
  }
  // the host side of the copy is memory-bound row work on up to a few hundred MB: a handful of threads, each taking every T-th plane as it arrives
  int T = 4;
  if (const char *e = getenv("JMHIP_HOST_THREADS")) T = atoi(e);
  T = std::max(1, std::min(T, std::min(planes, 16)));
  std::atomic<int> failed{0};
  auto work = [&](int t) {
    if (t && hipSetDevice(c->cfg.device) != hipSuccess) { failed = 1; return; }
    for (int p = t; p < planes; p += T) {
      if (!out_rows[(size_t)p * rows]) continue;
      if (hipEventSynchronize(c->pin_evt[p]) != hipSuccess) { failed = 1; return; }
      for (int j = 0; j < rows; j++) {
        const uint8_t *__restrict__ s = c->pin_host + p * plane + (size_t)j * width;
        if (pel_bytes == 1) memcpy(out_rows[(size_t)p * rows + j], s, (size_t)width);
        else {
          uint16_t *__restrict__ d = static_cast<uint16_t *>(out_rows[(size_t)p * rows + j]);
          for (int i = 0; i < width; i++) d[i] = s[i];
        }
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < T; t++) pool.emplace_back(work, t);
  work(0);
  for (auto &th : pool) th.join();
  if (failed) return jm_fail(c, JMHIP_ERR_DEVICE, "download of planes into rows");
  return JMHIP_OK;
}

extern "C" int jmhip_ref_download_luma_rows(jmhip_ctx *c, int ref, void *const *rows, int pel_bytes)
{
  if (!c) return JMHIP_ERR_ARG;
  if (ref < 0 || ref >= (int)c->refs.size()) return jm_fail(c, JMHIP_ERR_ARG, "ref slot out of range");
  if (!c->refs[ref].has_luma_sub) return jm_fail(c, JMHIP_ERR_ARG, "luma sub-pel planes not built (call jmhip_interp_luma)");
  return download_planes_rows(c, c->refs[ref].luma_sub, 16, c->Hp, c->Wp, rows, pel_bytes);
}

extern "C" int jmhip_ref_download_chroma_rows(jmhip_ctx *c, int ref, int uv, void *const *rows, int pel_bytes)
{
  if (!c) return JMHIP_ERR_ARG;
  if (ref < 0 || ref >= (int)c->refs.size() || uv < 0 || uv > 1) return jm_fail(c, JMHIP_ERR_ARG, "ref/uv out of range");
  if (!c->Wc) return jm_fail(c, JMHIP_ERR_ARG, "4:0:0 has no chroma");
  if (!c->refs[ref].has_cr_sub) return jm_fail(c, JMHIP_ERR_ARG, "chroma sub-pel planes not built (call jmhip_interp_chroma)");
  return download_planes_rows(c, c->refs[ref].cr_sub[uv], c->cg.sub_x * c->cg.sub_y, c->Hcp, c->Wcp, rows, pel_bytes);
}

extern "C" int jmhip_interp_luma(jmhip_ctx *c, int ref)
{
  if (!c) return JMHIP_ERR_ARG;
  if (ref < 0 || ref >= (int)c->refs.size()) return jm_fail(c, JMHIP_ERR_ARG, "ref slot out of range");
  if (!c->refs[ref].has_pic) return jm_fail(c, JMHIP_ERR_ARG, "reference picture not uploaded");
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  jm_stage_begin(c, JMHIP_STAGE_INTERP_LUMA);
  int rc = jm_launch_interp_luma(c, ref);
  jm_stage_end(c, JMHIP_STAGE_INTERP_LUMA);
  if (rc == JMHIP_OK) c->refs[ref].has_luma_sub = true;
  return rc;
}

// Sub-pel planes for the luma rows [row0, row1) of the PICTURE only (plus whatever the tile granularity adds): a rank that
// searches a band of macroblock rows needs the planes of that band +- (search range + predictor reach + block height) and
// nothing else. The caller owns that margin; rows outside keep whatever an earlier call left there.
static int interp_rows(jmhip_ctx *c, int ref, int row0, int row1, bool chroma);
extern "C" int jmhip_interp_rows(jmhip_ctx *c, int ref, int row0, int row1) { return c ? interp_rows(c, ref, row0, row1, true) : JMHIP_ERR_ARG; }
extern "C" int jmhip_interp_luma_rows(jmhip_ctx *c, int ref, int row0, int row1) { return c ? interp_rows(c, ref, row0, row1, false) : JMHIP_ERR_ARG; }

static int interp_rows(jmhip_ctx *c, int ref, int row0, int row1, bool chroma)
{
  if (ref < 0 || ref >= (int)c->refs.size()) return jm_fail(c, JMHIP_ERR_ARG, "ref slot out of range");
  if (!c->refs[ref].has_pic) return jm_fail(c, JMHIP_ERR_ARG, "reference picture not uploaded");
  if (row1 <= row0) return jm_fail(c, JMHIP_ERR_ARG, "jmhip_interp_rows: empty row range");
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  jm_stage_begin(c, JMHIP_STAGE_INTERP_LUMA);
  int rc = jm_launch_interp_luma(c, ref, row0 + JMHIP_PAD, row1 + JMHIP_PAD);
  jm_stage_end(c, JMHIP_STAGE_INTERP_LUMA);
  if (rc) return rc;
  c->refs[ref].has_luma_sub = true;
  if (c->Wc && chroma) {
    const int sy = c->cfg.yuv_format == JMHIP_YUV420 ? 1 : 0;       // luma rows -> chroma rows
    jm_stage_begin(c, JMHIP_STAGE_INTERP_CHROMA);
    rc = jm_launch_interp_chroma(c, ref, (row0 >> sy) + c->cg.pad_y - 1, ((row1 + sy) >> sy) + c->cg.pad_y + 1);
    jm_stage_end(c, JMHIP_STAGE_INTERP_CHROMA);
    if (rc) return rc;
    c->refs[ref].has_cr_sub = true;
  }
  return JMHIP_OK;
}

extern "C" int jmhip_interp_chroma(jmhip_ctx *c, int ref)
{
  if (!c) return JMHIP_ERR_ARG;
  if (ref < 0 || ref >= (int)c->refs.size()) return jm_fail(c, JMHIP_ERR_ARG, "ref slot out of range");
  if (!c->Wc) return jm_fail(c, JMHIP_ERR_ARG, "4:0:0 has no chroma");
  if (!c->refs[ref].has_pic) return jm_fail(c, JMHIP_ERR_ARG, "reference picture not uploaded");
  JM_HIP_CHECK(c, hipSetDevice(c->cfg.device));
  jm_stage_begin(c, JMHIP_STAGE_INTERP_CHROMA);
  int rc = jm_launch_interp_chroma(c, ref);
  jm_stage_end(c, JMHIP_STAGE_INTERP_CHROMA);
  if (rc == JMHIP_OK) c->refs[ref].has_cr_sub = true;
  return rc;
}
