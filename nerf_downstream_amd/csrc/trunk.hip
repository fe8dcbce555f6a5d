// Whole residual blocks as one host call (include/mink_hip.h, "whole residual blocks").
//
// Nothing here launches a kernel of its own: each function sequences the per-operator entry points of this
// library (mink_conv_* / mink_bn_* / mink_eltwise) over up to three HIP streams, with the cross-stream
// dependencies expressed through a small pool of library-owned events.  The point is the HOST: a Mink-ResNet14
// training step is ~300 launches; issued from a Python autograd graph one operator at a time they cost ~20 us
// each (4 ms per step, as much as the GPU needs for the step); issued from here they cost the ~3.5 us of
// hipLaunchKernel.
#include <algorithm>
#include <vector>

#include "common.h"

namespace mink {
namespace {

struct Scratch {  // bump allocator over one stream's scratch buffer
  char *base;
  int64_t bytes, off = 0;
  Scratch(void *p, int64_t n) : base((char *)p), bytes(n) {}
  void *take(int64_t n) {
    n = align_up(n > 0 ? n : 1, 256);
    if (!base || off + n > bytes) return nullptr;
    void *p = base + off;
    off += n;
    return p;
  }
};

constexpr int kEvents = 8;
hipEvent_t g_ev[kEvents];
bool g_ev_ready = false;

int ensure_events() {
  if (g_ev_ready) return MINK_OK;
  for (int i = 0; i < kEvents; ++i) MINK_HIP(hipEventCreateWithFlags(&g_ev[i], hipEventDisableTiming));
  g_ev_ready = true;
  return MINK_OK;
}

// `waiter` continues only after everything queued on `producer` so far (no-op for one and the same stream)
int order_after(hipStream_t waiter, hipStream_t producer, int slot) {
  if (waiter == producer) return MINK_OK;
  MINK_HIP(hipEventRecord(g_ev[slot], producer));
  MINK_HIP(hipStreamWaitEvent(waiter, g_ev[slot], 0));
  return MINK_OK;
}

MinkBlockDoneHook g_block_done_hook = nullptr;  // data parallelism: the caller issues a block's collective from ex->wgrad (mink_net_backward)
MinkStageHook g_stage_hook = nullptr;  // test / timeline instrumentation between the stages of mink_net_*

#define TRY(expr)        \
  do {                   \
    int rc_ = (expr);    \
    if (rc_) return rc_; \
  } while (0)

struct ConvOut {
  float *slabs = nullptr;
  int32_t nslab = 0;           // > 1: `slabs` holds that many partial sums of y (few-row layers)
  bool small = false;
  const double *partial = nullptr;  // column (sum, sum of squares) partials of y, `rows` of them (0: mean / invstd are final)
  int32_t rows = 0;
};

// y = conv(x) and the column partials of y for the batch norm that follows (from the convolution's own epilogue or split-K
// reduce when the launch shape allows: co.rows > 0, finalized by the norm's apply pass; else mean / invstd are computed here)
int conv_stats(const MinkConvLayer &c, const MinkNormLayer &nm, const float *x, int64_t n_in, int64_t n_out, float *y, void *ws_base,
               int64_t ws_bytes, hipStream_t st, ConvOut &co) {
  Scratch ws(ws_base, ws_bytes);
  const int ksplit = mink_conv_plan(n_out, c.K, c.cin, c.cout, 0);
  const int64_t slab_bytes = ksplit > 1 ? 4ll * ksplit * n_out * c.cout : 0, stats_bytes = mink_conv_stats_workspace_bytes(n_out, c.cout);
  float *slabs = ksplit > 1 ? (float *)ws.take(slab_bytes) : nullptr;
  void *stats_ws = ws.take(stats_bytes);
  double *partial = (double *)ws.take(512ll * 2 * c.cout * sizeof(double));
  MINK_REQUIRE((ksplit == 1 || slabs) && stats_ws && partial, "block: scratch too small for a %lld x %d convolution",
               (long long)n_out, c.cout);
  int32_t rows = 0;
  TRY(mink_conv_gather_gemm_stats(x, n_in, c.cin, c.cin, c.w, c.nbr, n_out, c.K, y, c.cout, c.cout, nullptr, ksplit, slabs, slab_bytes,
                                  partial, &rows, stats_ws, stats_bytes, st));
  co.partial = partial, co.rows = rows;
  if (rows > 0) return MINK_OK;
  const float mom = nm.running_mean ? nm.momentum : 0.f;
  const int64_t bn_bytes = mink_bn_workspace_bytes(n_out, c.cout);
  void *bn_ws = ws.take(bn_bytes);
  MINK_REQUIRE(bn_ws, "block: scratch too small for batch-norm statistics");
  return mink_bn_stats(y, n_out, c.cout, nm.eps, mom, nm.mean, nm.invstd, nm.running_mean, nm.running_var, bn_ws, bn_bytes, st);
}

// Few-row layers (mink_bn_small_rows): the convolution leaves its split-K slabs and ONE launch sums them, takes the
// statistics and applies norm + residual + ReLU -- two launches per convolution + norm instead of four.
// ... where that one launch is faster than the three it replaces: its time grows with rows x slabs (a workgroup owns 8 channels
// and ALL rows; measured 7 us at 128 x 2, 13 us at 128 x 14 and 532 x 5, 22 us at 512 x 14 against ~22 us for the three launches)
bool small_layer(int64_t n_out, int cout, int nslab = 1) {
  return n_out <= mink_bn_small_rows() && cout % 16 == 0 && n_out * (nslab > 1 ? nslab : 1) <= 3000;
}

// the convolution of a conv + norm pair (and, on the general path, the statistics and their finalize)
int conv_for_norm(const MinkConvLayer &c, const MinkNormLayer &nm, const float *x, int64_t n_in, int64_t n_out, float *y, void *ws_base,
                  int64_t ws_bytes, hipStream_t st, ConvOut &co) {
  const int ksplit = mink_conv_plan(n_out, c.K, c.cin, c.cout, 0);
  co.small = small_layer(n_out, c.cout, ksplit);
  if (!co.small) return conv_stats(c, nm, x, n_in, n_out, y, ws_base, ws_bytes, st, co);
  Scratch ws(ws_base, ws_bytes);
  const int64_t slab_bytes = ksplit > 1 ? 4ll * ksplit * n_out * c.cout : 0;
  co.slabs = ksplit > 1 ? (float *)ws.take(slab_bytes) : nullptr;
  MINK_REQUIRE(ksplit == 1 || co.slabs, "block: scratch too small for a %lld x %d convolution", (long long)n_out, c.cout);
  return mink_conv_gather_gemm_slabs(x, n_in, c.cin, c.cin, c.w, 0, 0, c.nbr, n_out, c.K, nullptr, 0, y, c.cout, c.cout, ksplit, co.slabs,
                                     slab_bytes, &co.nslab, st);
}

// out = act(norm(y) [+ residual]) behind conv_for_norm
int norm_act(const MinkNormLayer &nm, int64_t n_out, int C, float *y, const float *residual, int relu, float *out, const ConvOut &co,
             hipStream_t st) {
  if (!co.small && co.rows > 0)  // (the statistics' finalize runs inside the apply pass when the partial rows are few)
    return mink_bn_apply_from_partials(y, n_out, C, co.partial, co.rows, nm.eps, nm.running_mean ? nm.momentum : 0.f, nm.gamma, nm.beta, residual,
                                       relu, out, nm.mean, nm.invstd, nm.running_mean, nm.running_var, st);
  if (!co.small) return mink_bn_apply(y, n_out, C, nm.mean, nm.invstd, nm.gamma, nm.beta, residual, relu, out, st);
  return mink_bn_small_fwd(co.slabs, co.nslab > 1 ? co.nslab : 0, n_out, C, y, nm.eps, nm.running_mean ? nm.momentum : 0.f, nm.gamma, nm.beta,
                           residual, relu, out, nm.mean, nm.invstd, nm.running_mean, nm.running_var, st);
}

struct Lane {  // a stream and what is left of its scratch buffer
  hipStream_t st;
  void *ws;
  int64_t bytes;
  Lane after(int64_t used) const { return Lane{st, (char *)ws + align_up(used, 256), bytes - align_up(used, 256)}; }
};

// gy (complete on data.st) -> dW on weight.st, dX (optional) on data.st
// (ev_slot < 0: the caller has already ordered weight.st after gy)
// gx_slabs (stride-1 convolutions only): leave the data gradient as split-K slabs for the caller's next kernel
// (mink_bn_small_bwd sums them); gx_slabs->nslab <= 1 on return means gx is complete as usual
int conv_backward(const MinkConvLayer &c, const float *x, int64_t n_in, int64_t n_out, const float *gy, float *gx, Lane data,
                  Lane weight, int ev_slot, ConvOut *gx_slabs = nullptr) {
  if (ev_slot >= 0) TRY(order_after(weight.st, data.st, ev_slot));
  {
    Scratch ws(weight.ws, weight.bytes);
    const int64_t need = mink_conv_wgrad_workspace_bytes(n_out, c.K, c.cin, c.cout);
    void *slabs = need > 0 ? ws.take(need) : nullptr;
    MINK_REQUIRE(need == 0 || slabs, "block: weight-gradient scratch too small");
    TRY(mink_conv_wgrad(x, n_in, c.cin, c.cin, gy, c.cout, c.cout, c.nbr, n_out, c.K, c.dw, slabs, need, weight.st));
  }
  if (!gx) return MINK_OK;
  Scratch ws(data.ws, data.bytes);
  if (c.stride == 1) {  // centred odd kernel: the transposed table is the table with the offsets flipped
    const int ksplit = mink_conv_plan(n_out, c.K, c.cout, c.cin, 0);
    const int64_t slab_bytes = ksplit > 1 ? 4ll * ksplit * n_out * c.cin : 0;
    float *slabs = ksplit > 1 ? (float *)ws.take(slab_bytes) : nullptr;
    MINK_REQUIRE(ksplit == 1 || slabs, "block: data-gradient scratch too small");
    if (gx_slabs) {
      gx_slabs->slabs = slabs;
      return mink_conv_gather_gemm_slabs(gy, n_out, c.cout, c.cout, c.w, 1, 1, c.nbr, n_out, c.K, nullptr, 0, gx, c.cin, c.cin, ksplit,
                                         slabs, slab_bytes, &gx_slabs->nslab, data.st);
    }
    return mink_conv_gather_gemm(gy, n_out, c.cout, c.cout, c.w, 1, 1, c.nbr, n_out, c.K, nullptr, 0, gx, c.cin, c.cin, nullptr,
                                 ksplit, slabs, slab_bytes, data.st);
  }
  MINK_REQUIRE(c.nbr_t, "block: a strided convolution needs its transposed table for the data gradient");
  const int64_t rows = c.perm ? c.n_perm : n_in;
  const int ksplit = mink_conv_plan(rows, c.K, c.cout, c.cin, c.perm ? 1 : 0);
  const int64_t slab_bytes = ksplit > 1 ? 4ll * ksplit * n_in * c.cin : 0;
  float *slabs = ksplit > 1 ? (float *)ws.take(slab_bytes) : nullptr;
  MINK_REQUIRE(ksplit == 1 || slabs, "block: data-gradient scratch too small");
  return mink_conv_gather_gemm(gy, n_out, c.cout, c.cout, c.w, 1, 0, c.nbr_t, n_in, c.K, c.perm, c.perm ? c.n_perm : 0, gx, c.cin,
                               c.cin, nullptr, ksplit, slabs, slab_bytes, data.st);
}

int check_conv(const MinkConvLayer &c, const char *what, bool backward) {
  MINK_REQUIRE(c.w && c.nbr && c.K >= 1 && c.K <= 27 && c.cin >= 4 && c.cout >= 4 && (c.cin & 3) == 0 && (c.cout & 3) == 0,
               "%s: bad convolution layer (K=%d, %d -> %d)", what, c.K, c.cin, c.cout);
  MINK_REQUIRE(!backward || c.dw, "%s: NULL weight-gradient buffer", what);
  return MINK_OK;
}

int check_norm(const MinkNormLayer &n, const char *what, bool backward) {
  MINK_REQUIRE(n.gamma && n.beta && n.mean && n.invstd && (n.running_mean == nullptr) == (n.running_var == nullptr),
               "%s: bad batch-norm layer", what);
  MINK_REQUIRE(!backward || (n.dgamma && n.dbeta), "%s: NULL batch-norm gradient buffer", what);
  return MINK_OK;
}

}  // namespace
}  // namespace mink

using namespace mink;

extern "C" {

int mink_stream_create_cu_subset(int32_t first_cu, int32_t n_cus, int32_t total_cus, void **stream_out) {
  MINK_REQUIRE(stream_out && total_cus >= 1 && total_cus <= 1024 && first_cu >= 0 && n_cus >= 1 && first_cu + n_cus <= total_cus,
               "stream_create_cu_subset: bad compute-unit range %d + %d of %d", first_cu, n_cus, total_cus);
  uint32_t mask[32] = {0};
  for (int i = first_cu; i < first_cu + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t st = nullptr;
  MINK_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)((total_cus + 31) / 32), mask));
  *stream_out = (void *)st;
  return MINK_OK;
}

int mink_stream_destroy(void *stream) {
  if (stream) MINK_HIP(hipStreamDestroy((hipStream_t)stream));
  return MINK_OK;
}

int64_t mink_block_workspace_bytes(int64_t n_in, int64_t n_out, int32_t cin, int32_t cout) {
  // generous upper bound of what any single operator of the block carves from one stream's scratch:
  // split-K slabs (the planner keeps them under 128 MiB) + statistics partials + weight-gradient slabs
  const int64_t rows = n_in > n_out ? n_in : n_out;
  const int64_t cmax = cin > cout ? cin : cout;
  int64_t b = (160ll << 20);
  b += mink_conv_stats_workspace_bytes(rows, cmax) + 512ll * 2 * cmax * 8 + mink_bn_workspace_bytes(rows, cmax);
  int64_t wg = 0;
  for (int k : {1, 27}) {
    wg = std::max<int64_t>(wg, mink_conv_wgrad_workspace_bytes(n_out, k, cin, cout));
    wg = std::max<int64_t>(wg, mink_conv_wgrad_workspace_bytes(n_out, k, cout, cout));
  }
  return align_up(b + wg + 4096, 256);
}

int64_t mink_block_grad_scratch_floats(int64_t n_in, int64_t n_out, int32_t cin, int32_t cout, int32_t has_down) {
  // g_y2, g_res, g_h1, g_y1 [n_out][cout]; identity shortcut: g_xa [n_in][cin]; with a down path: g_yd [n_out][cout] and
  // g_sc [n_out][cin] (the shortcut's data gradient per output row)
  return (4 + (has_down ? 1 : 0)) * n_out * cout + (has_down ? n_out : n_in) * cin + 64;
}

int mink_stem_supported(int64_t n, int32_t cin, int32_t cout, int32_t K) {
  return n > 0 && (cin & 3) == 0 && (cout & 3) == 0 && mink_conv_wgrad_bn_relu_pool_supported(n, cin, cin, n, K, cout);
}

int mink_stem_forward(const MinkStem *s, const MinkExec *ex) {
  MINK_REQUIRE(s && ex, "stem_forward: NULL descriptor");
  TRY(check_conv(s->conv, "stem_forward", false));
  TRY(check_norm(s->norm, "stem_forward", false));
  MINK_REQUIRE(s->x && s->y && s->out && s->nbr_pool && s->n >= 1 && s->n_pool >= 1, "stem_forward: bad arguments");
  hipStream_t st = (hipStream_t)ex->compute;
  if (s->xb) {  // bf16 storage of the full-resolution stage: x copy and conv output in bf16, statistics of the stored values
    const MinkConvLayer &c = s->conv;
    MINK_REQUIRE(mink_stem_conv_bf16s_supported(s->n, s->n, c.K, c.cin, c.cout), "stem_forward: bf16 storage does not support this stem");
    Scratch ws(ex->ws_compute, ex->ws_bytes);
    const int rows = mink_stem_conv_bf16s_stats_rows();
    double *partial = (double *)ws.take((int64_t)rows * 2 * c.cout * sizeof(double));
    MINK_REQUIRE(partial, "stem_forward: scratch too small");
    if (!s->xb_ready) TRY(mink_rows_to_bf16(s->x, s->n, c.cin, c.cin, s->xb, st));
    TRY(mink_stem_conv_bf16s(s->xb, s->n, c.w, c.cin, c.nbr, s->n, c.K, s->y, c.cout, partial, rows, st));
    TRY(mink_bn_stats_from_partials(partial, rows, s->n, c.cout, s->norm.eps, s->norm.running_mean ? s->norm.momentum : 0.f, s->norm.mean,
                                    s->norm.invstd, s->norm.running_mean, s->norm.running_var, st));
    return mink_bn_relu_pool_fwd_b16(s->y, c.cout, s->norm.mean, s->norm.invstd, s->norm.gamma, s->norm.beta, s->nbr_pool, s->n_pool, 8,
                                     s->out, st);
  }
  ConvOut co;
  TRY(conv_stats(s->conv, s->norm, s->x, s->n, s->n, s->y, ex->ws_compute, ex->ws_bytes, st, co));
  if (co.rows > 0)
    TRY(mink_bn_stats_from_partials(co.partial, co.rows, s->n, s->conv.cout, s->norm.eps, s->norm.running_mean ? s->norm.momentum : 0.f,
                                    s->norm.mean, s->norm.invstd, s->norm.running_mean, s->norm.running_var, st));
  return mink_bn_relu_pool_fwd(s->y, s->conv.cout, s->norm.mean, s->norm.invstd, s->norm.gamma, s->norm.beta, s->nbr_pool,
                               s->n_pool, 8, s->out, st);
}

int mink_stem_backward(const MinkStem *s, const MinkExec *ex) {
  MINK_REQUIRE(s && ex, "stem_backward: NULL descriptor");
  TRY(check_conv(s->conv, "stem_backward", true));
  TRY(check_norm(s->norm, "stem_backward", true));
  MINK_REQUIRE(s->x && s->y && s->g_out && s->in2out && s->n >= 1 && s->n_pool >= 1, "stem_backward: bad arguments");
  hipStream_t st = (hipStream_t)ex->compute;
  Scratch ws(ex->ws_compute, ex->ws_bytes);
  void *bn_ws = ws.take(mink_bn_workspace_bytes(s->n, s->conv.cout));
  const int64_t need = mink_conv_wgrad_workspace_bytes(s->n, s->conv.K, s->conv.cin, s->conv.cout);
  void *slabs = need > 0 ? ws.take(need) : nullptr;
  MINK_REQUIRE(bn_ws && (need == 0 || slabs), "stem_backward: scratch too small");
  if (s->xb) {
    TRY(mink_bn_relu_pool_bwd_b16(s->g_out, s->y, s->n, s->conv.cout, s->norm.mean, s->norm.invstd, s->norm.gamma, s->norm.beta, s->in2out,
                                  s->norm.dgamma, s->norm.dbeta, bn_ws, mink_bn_workspace_bytes(s->n, s->conv.cout), st));
    return mink_conv_wgrad_bn_relu_pool_b16(s->xb, s->n, s->conv.cin, s->y, s->conv.cout, s->g_out, s->n_pool, s->in2out, s->norm.mean,
                                            s->norm.invstd, s->norm.gamma, s->norm.beta, s->norm.dgamma, s->norm.dbeta, s->conv.nbr,
                                            s->n, s->conv.K, s->conv.dw, slabs, need, st);
  }
  TRY(mink_bn_relu_pool_bwd(s->g_out, s->y, s->n, s->conv.cout, s->norm.mean, s->norm.invstd, s->norm.gamma, s->norm.beta,
                            s->in2out, nullptr, s->norm.dgamma, s->norm.dbeta, bn_ws, mink_bn_workspace_bytes(s->n, s->conv.cout), st));
  // the gradient w.r.t. the convolution output is recomputed inside the weight-gradient kernel's operand load
  return mink_conv_wgrad_bn_relu_pool(s->x, s->n, s->conv.cin, s->conv.cin, s->y, s->conv.cout, s->g_out, s->n_pool, s->in2out,
                                      s->norm.mean, s->norm.invstd, s->norm.gamma, s->norm.beta, s->norm.dgamma,
                                      s->norm.dbeta, s->conv.nbr, s->n, s->conv.K, s->conv.dw, slabs, need, st);
}

int mink_block_forward(const MinkBasicBlock *b, const MinkExec *ex) {
  MINK_REQUIRE(b && ex, "block_forward: NULL descriptor");
  TRY(check_conv(b->conv1, "block_forward conv1", false));
  TRY(check_conv(b->conv2, "block_forward conv2", false));
  TRY(check_norm(b->norm1, "block_forward norm1", false));
  TRY(check_norm(b->norm2, "block_forward norm2", false));
  const bool down = b->down.w != nullptr;
  const int C = b->conv1.cout;
  MINK_REQUIRE(b->x && b->y1 && b->h1 && b->y2 && b->out && b->n_in >= 1 && b->n_out >= 1 && b->conv2.cin == C &&
                   b->conv2.cout == C,
               "block_forward: bad arguments");
  MINK_REQUIRE(down || (b->n_in == b->n_out && b->conv1.cin == C), "block_forward: an identity shortcut needs equal shapes");
  TRY(ensure_events());
  hipStream_t st = (hipStream_t)ex->compute, br = (hipStream_t)ex->branch;
  const float *shortcut = b->x;
  if (down) {  // shortcut branch: 1x1x1 strided convolution + norm beside conv1 / norm1 / conv2
    TRY(check_conv(b->down, "block_forward downsample", false));
    TRY(check_norm(b->normd, "block_forward downsample norm", false));
    MINK_REQUIRE(b->yd && b->sd && b->down.cout == C && b->down.cin == b->conv1.cin, "block_forward: bad downsample path");
    TRY(order_after(br, st, 0));  // x is ready
    ConvOut cd;
    TRY(conv_for_norm(b->down, b->normd, b->x, b->n_in, b->n_out, b->yd, br == st ? ex->ws_compute : ex->ws_branch, ex->ws_bytes, br, cd));
    TRY(norm_act(b->normd, b->n_out, C, b->yd, nullptr, 0, b->sd, cd, br));
    shortcut = b->sd;
  }
  ConvOut c1, c2;
  TRY(conv_for_norm(b->conv1, b->norm1, b->x, b->n_in, b->n_out, b->y1, ex->ws_compute, ex->ws_bytes, st, c1));
  TRY(norm_act(b->norm1, b->n_out, C, b->y1, nullptr, 1, b->h1, c1, st));
  TRY(conv_for_norm(b->conv2, b->norm2, b->h1, b->n_out, b->n_out, b->y2, ex->ws_compute, ex->ws_bytes, st, c2));
  if (down) TRY(order_after(st, br, 1));
  return norm_act(b->norm2, b->n_out, C, b->y2, shortcut, 1, b->out, c2, st);
}

}  // extern "C"

namespace {
// A block's input gradient that has not been summed yet: g = slab 0 + ... + slab nslab-1 + addend, to be written to `sum` by
// whoever consumes it -- the batch-norm backward of the block BEFORE (mink_bn_bwd_slabs / mink_bn_small_bwd), which reads the
// gradient anyway.  An identity block then ends with its conv1 data gradient's kernel: no reduce launch, no add launch.
struct PendingGrad {
  const float *slabs = nullptr;
  int32_t nslab = 0;
  const float *addend = nullptr;
};

// in : this block's g_out arrives as `in` (nslab > 1) and is written to b->g_out by the first kernel here
// out: an identity block may leave its input gradient pending (out->nslab > 1; the caller's next block consumes it)
int block_backward_impl(const MinkBasicBlock *b, const MinkExec *ex, const PendingGrad *in, PendingGrad *out) {
  if (out) *out = PendingGrad{};
  MINK_REQUIRE(b && ex, "block_backward: NULL descriptor");
  TRY(check_conv(b->conv1, "block_backward conv1", true));
  TRY(check_conv(b->conv2, "block_backward conv2", true));
  TRY(check_norm(b->norm1, "block_backward norm1", true));
  TRY(check_norm(b->norm2, "block_backward norm2", true));
  const bool down = b->down.w != nullptr;
  const int C = b->conv1.cout, cin = b->conv1.cin;
  MINK_REQUIRE(b->x && b->y1 && b->h1 && b->y2 && b->out && b->g_out && b->g_tmp && b->n_in >= 1 && b->n_out >= 1,
               "block_backward: bad arguments");
  TRY(ensure_events());
  hipStream_t st = (hipStream_t)ex->compute, br = (hipStream_t)ex->branch, wst = (hipStream_t)ex->wgrad;
  const Lane compute{st, ex->ws_compute, ex->ws_bytes};
  const Lane branch{br, br == st ? ex->ws_compute : ex->ws_branch, ex->ws_bytes};
  const Lane weight{wst, wst == st ? ex->ws_compute : ex->ws_wgrad, ex->ws_bytes};
  const int64_t no = b->n_out * C, ni = b->n_in * cin;
  float *g_y2 = b->g_tmp, *g_res = g_y2 + no, *g_h1 = g_res + no, *g_y1 = g_h1 + no;
  float *g_xa = g_y1 + no;  // identity shortcut only (then g_x = g_xa + g_res)
  float *g_yd = g_y1 + no;  // down path only
  const int64_t bn_bytes = mink_bn_workspace_bytes(b->n_out, C);
  MINK_REQUIRE(ex->ws_compute && ex->ws_bytes > 2 * bn_bytes + 256, "block_backward: scratch too small");
  const bool want_gx = b->g_x != nullptr;
  const bool small = small_layer(b->n_out, C);
  // out = relu(norm2(y2) + shortcut)
  const bool pend = in && in->nslab > 1;  // g_out = the pending slabs + addend of the block behind, summed here into b->g_out
  float *g_out_w = const_cast<float *>(b->g_out);
  if (pend && small_layer(b->n_out, C, in->nslab))
    TRY(mink_bn_small_bwd(in->slabs, in->nslab, in->addend, g_out_w, b->y2, b->out, b->n_out, C, b->norm2.mean, b->norm2.invstd,
                          b->norm2.gamma, 1, g_y2, g_res, b->norm2.dgamma, b->norm2.dbeta, st));
  else if (pend)
    TRY(mink_bn_bwd_slabs(in->slabs, in->nslab, in->addend, g_out_w, b->y2, b->out, b->n_out, C, b->norm2.mean, b->norm2.invstd, b->norm2.gamma,
                          1, g_y2, g_res, b->norm2.dgamma, b->norm2.dbeta, compute.ws, bn_bytes, st));
  else if (small)
    TRY(mink_bn_small_bwd(b->g_out, 0, nullptr, nullptr, b->y2, b->out, b->n_out, C, b->norm2.mean, b->norm2.invstd, b->norm2.gamma, 1, g_y2,
                          g_res, b->norm2.dgamma, b->norm2.dbeta, st));
  else
    TRY(mink_bn_bwd(b->g_out, b->y2, b->out, b->n_out, C, b->norm2.mean, b->norm2.invstd, b->norm2.gamma, 1, g_y2, g_res,
                    b->norm2.dgamma, b->norm2.dbeta, compute.ws, bn_bytes, st));
  // ONE event on the compute stream hands g_y2 / g_res to both auxiliary streams (every event record or wait on the
  // compute stream is a barrier packet in the chain of small dependent kernels)
  const bool aux = wst != st || (down && br != st);
  if (aux) MINK_HIP(hipEventRecord(g_ev[2], st));
  if (wst != st) MINK_HIP(hipStreamWaitEvent(wst, g_ev[2], 0));
  float *g_sc = nullptr;  // [n_out][cin]: the shortcut's data gradient per OUTPUT row (dense), scattered into g_x at the end
  if (down) {  // shortcut branch beside the main one: its norm, its weight gradient, its data gradient
    TRY(check_conv(b->down, "block_backward downsample", true));
    TRY(check_norm(b->normd, "block_backward downsample norm", true));
    MINK_REQUIRE(b->yd && branch.ws, "block_backward: bad downsample path");
    MINK_REQUIRE(b->down.K == 1, "block_backward: the shortcut convolution must have kernel volume 1");
    if (br != st) MINK_HIP(hipStreamWaitEvent(br, g_ev[2], 0));
    // (on one stream the branch shares the compute scratch: its batch-norm partials sit behind the main chain's)
    const Lane bl = br == st ? branch.after(bn_bytes) : branch;
    if (small)
      TRY(mink_bn_small_bwd(g_res, 0, nullptr, nullptr, b->yd, nullptr, b->n_out, C, b->normd.mean, b->normd.invstd, b->normd.gamma, 0, g_yd, nullptr,
                            b->normd.dgamma, b->normd.dbeta, br));
    else
      TRY(mink_bn_bwd(g_res, b->yd, nullptr, b->n_out, C, b->normd.mean, b->normd.invstd, b->normd.gamma, 0, g_yd, nullptr,
                      b->normd.dgamma, b->normd.dbeta, bl.ws, bn_bytes, br));
    const Lane rest_b = bl.after(bn_bytes);
    if (want_gx) {
      // The shortcut convolution (kernel volume 1, stride 2) reaches only the input voxels that sit on an output
      // coordinate -- about one row in eight.  Its data gradient is a plain GEMM over the OUTPUT rows, computed here
      // beside the main branch; what is left on the compute stream is adding those rows into g_x (below).
      g_sc = g_yd + no;  // (in the caller's gradient scratch: stream scratch is recycled by the launches that follow)
      TRY(mink_dense_xwt(g_yd, b->down.w, b->n_out, C, cin, g_sc, br));
      if (br != st) MINK_HIP(hipEventRecord(g_ev[6], br));
    }
    TRY(conv_backward(b->down, b->x, b->n_in, b->n_out, g_yd, nullptr, rest_b, weight, br == wst ? -1 : 3));
  }
  const Lane rest = compute.after(2 * bn_bytes);
  {  // conv2's data gradient stays in its split-K slabs: norm1's backward sums them on the way in (no reduce launch)
    ConvOut gs;
    TRY(conv_backward(b->conv2, b->h1, b->n_out, b->n_out, g_y2, g_h1, rest, weight, -1, C <= 1024 ? &gs : nullptr));
    const bool sl = gs.nslab > 1;
    if (small_layer(b->n_out, C, gs.nslab))
      TRY(mink_bn_small_bwd(sl ? gs.slabs : g_h1, sl ? gs.nslab : 0, nullptr, sl ? g_h1 : nullptr, b->y1, b->h1, b->n_out, C, b->norm1.mean,
                            b->norm1.invstd, b->norm1.gamma, 1, g_y1, nullptr, b->norm1.dgamma, b->norm1.dbeta, st));
    else if (sl && C <= 1024)
      TRY(mink_bn_bwd_slabs(gs.slabs, gs.nslab, nullptr, g_h1, b->y1, b->h1, b->n_out, C, b->norm1.mean, b->norm1.invstd, b->norm1.gamma, 1, g_y1,
                            nullptr, b->norm1.dgamma, b->norm1.dbeta, compute.ws, bn_bytes, st));
    else {
      TRY(mink_bn_bwd(g_h1, b->y1, b->h1, b->n_out, C, b->norm1.mean, b->norm1.invstd, b->norm1.gamma, 1, g_y1, nullptr,
                      b->norm1.dgamma, b->norm1.dbeta, compute.ws, bn_bytes, st));
    }
  }
  // an identity block may hand its input gradient on UNSUMMED: conv1's data-gradient slabs + g_res (the caller's next block sums
  // them inside its own batch-norm backward; needs the same channel count there: the block before an identity block has it)
  ConvOut g1;
  const bool defer = out && want_gx && !down && b->conv1.stride == 1 && cin <= 1024;
  TRY(conv_backward(b->conv1, b->x, b->n_in, b->n_out, g_y1, want_gx ? (down ? b->g_x : g_xa) : nullptr, rest, weight, 5,
                    defer ? &g1 : nullptr));
  if (!want_gx) return MINK_OK;
  if (!down) {
    if (defer && g1.nslab > 1) {
      out->slabs = g1.slabs, out->nslab = g1.nslab, out->addend = g_res;
      return MINK_OK;
    }
    return mink_eltwise(g_xa, g_res, ni, 2, b->g_x, st);
  }
  if (br != st) MINK_HIP(hipStreamWaitEvent(st, g_ev[6], 0));
  return mink_rows_scatter_add(g_sc, b->down.nbr, b->n_out, cin, b->g_x, st);
}

}  // namespace

extern "C" {

int mink_block_backward(const MinkBasicBlock *b, const MinkExec *ex) { return block_backward_impl(b, ex, nullptr, nullptr); }

}  // extern "C"

// ------------------------------------------------------------------------------------------- the whole trunk
namespace {

constexpr int64_t kNetAlign = 64;  // floats: every block's region starts on a 256-byte boundary

struct BlockShape {
  int64_t n_in, n_out, act, stats, gtmp, gx;  // floats: activations, statistics, gradient scratch, input gradient
  int level_in, level_out;
  bool down;
};

int block_shape(const MinkNet *net, const MinkLevelMaps *levels, int32_t n_levels, int i, int &level, BlockShape &sh) {
  const MinkBasicBlock &b = net->blocks[i];
  const int stride = b.conv1.stride;
  MINK_REQUIRE(stride == 1 || stride == 2, "net: block %d has stride %d", i, stride);
  sh.level_in = level;
  if (stride == 2) ++level;
  sh.level_out = level;
  MINK_REQUIRE(level < n_levels, "net: block %d needs level %d, %d given", i, level, n_levels);
  sh.down = b.down.w != nullptr;
  sh.n_in = levels[sh.level_in].n, sh.n_out = levels[sh.level_out].n;
  MINK_REQUIRE(sh.n_in >= 1 && sh.n_out >= 1, "net: empty level at block %d", i);
  const int C = b.conv1.cout;
  sh.act = align_up((sh.down ? 6 : 4) * sh.n_out * C, kNetAlign);
  sh.stats = align_up(6 * C, kNetAlign);
  sh.gtmp = align_up(mink_block_grad_scratch_floats(sh.n_in, sh.n_out, b.conv1.cin, C, sh.down ? 1 : 0), kNetAlign);
  sh.gx = align_up(sh.n_in * b.conv1.cin, kNetAlign);
  return MINK_OK;
}

// points block i's per-step fields at its region of the activation arena and at this batch's maps
int bind_block(MinkBasicBlock &b, const BlockShape &sh, const MinkLevelMaps *levels, float *region, const float *x) {
  const MinkLevelMaps &lo = levels[sh.level_out];
  const int C = b.conv1.cout;
  const int64_t cnt = sh.n_out * C;
  b.n_in = sh.n_in, b.n_out = sh.n_out, b.x = x;
  b.y1 = region, b.h1 = region + cnt, b.y2 = region + 2 * cnt, b.out = region + 3 * cnt;
  b.yd = sh.down ? region + 4 * cnt : nullptr, b.sd = sh.down ? region + 5 * cnt : nullptr;
  float *st = region + sh.act;
  b.norm1.mean = st, b.norm1.invstd = st + C, b.norm2.mean = st + 2 * C, b.norm2.invstd = st + 3 * C;
  b.normd.mean = st + 4 * C, b.normd.invstd = st + 5 * C;
  b.conv2.nbr = lo.nbr3, b.conv2.nbr_t = nullptr, b.conv2.perm = nullptr, b.conv2.n_perm = 0;
  if (b.conv1.stride == 2) {
    b.conv1.nbr = lo.down3, b.conv1.nbr_t = lo.down3_t, b.conv1.perm = lo.perm, b.conv1.n_perm = lo.perm ? lo.n_perm : 0;
    MINK_REQUIRE(sh.down && lo.down1, "net: a strided block needs its shortcut convolution and the k=1 table");
    b.down.nbr = lo.down1, b.down.nbr_t = nullptr, b.down.perm = nullptr, b.down.n_perm = 0;
  } else {
    MINK_REQUIRE(!sh.down, "net: a stride-1 block with a shortcut convolution is not sequenced here");
    b.conv1.nbr = lo.nbr3, b.conv1.nbr_t = nullptr, b.conv1.perm = nullptr, b.conv1.n_perm = 0;
  }
  MINK_REQUIRE(b.conv1.nbr && b.conv2.nbr, "net: missing neighbour table at level %d", sh.level_out);
  return MINK_OK;
}

}  // namespace

extern "C" {

int mink_set_block_done_hook(MinkBlockDoneHook hook) {
  g_block_done_hook = hook;
  return MINK_OK;
}

int mink_set_stage_hook(MinkStageHook hook) {
  g_stage_hook = hook;
  return MINK_OK;
}

int mink_event_create(void **event_out) {
  MINK_REQUIRE(event_out, "event_create: NULL");
  hipEvent_t ev = nullptr;
  MINK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  *event_out = (void *)ev;
  return MINK_OK;
}

int mink_event_destroy(void *event) {
  if (event) MINK_HIP(hipEventDestroy((hipEvent_t)event));
  return MINK_OK;
}

int mink_stream_wait_event(void *stream, void *event) {
  MINK_REQUIRE(event, "stream_wait_event: NULL event");
  MINK_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return MINK_OK;
}

int mink_net_sizes(const MinkNet *net, const MinkLevelMaps *levels, int32_t n_levels, int64_t *act_floats, int64_t *grad_floats,
                   int64_t *ws_bytes) {
  MINK_REQUIRE(net && levels && n_levels >= 1 && net->n_blocks >= 1 && net->blocks, "net_sizes: bad arguments");
  int64_t act = 0, grad = 0, ws = 0;
  int level = 0;
  for (int i = 0; i < net->n_blocks; ++i) {
    BlockShape sh;
    TRY(block_shape(net, levels, n_levels, i, level, sh));
    act += sh.act + sh.stats, grad += sh.gtmp + sh.gx;
    ws = std::max(ws, mink_block_workspace_bytes(sh.n_in, sh.n_out, net->blocks[i].conv1.cin, net->blocks[i].conv1.cout));
  }
  if (act_floats) *act_floats = act;
  if (grad_floats) *grad_floats = grad;
  if (ws_bytes) *ws_bytes = ws;
  return MINK_OK;
}

int mink_net_forward(MinkNet *net, const MinkLevelMaps *levels, int32_t n_levels, float *arena, int64_t arena_floats,
                     const MinkExec *ex) {
  MINK_REQUIRE(net && levels && arena && ex && net->blocks && net->n_blocks >= 1, "net_forward: bad arguments");
  if (g_stage_hook) g_stage_hook(-1, 0);
  if (net->with_stem) TRY(mink_stem_forward(&net->stem, ex));
  MINK_REQUIRE(net->stem.out && net->stem.n_pool == levels[0].n, "net_forward: the stem output must be level 0 (%lld rows, %lld given)",
               (long long)net->stem.n_pool, (long long)levels[0].n);
  const float *x = net->stem.out;
  int level = 0;
  int64_t off = 0;
  for (int i = 0; i < net->n_blocks; ++i) {
    BlockShape sh;
    TRY(block_shape(net, levels, n_levels, i, level, sh));
    MINK_REQUIRE(off + sh.act + sh.stats <= arena_floats, "net_forward: activation arena of %lld floats is too small (mink_net_sizes)",
                 (long long)arena_floats);
    MinkBasicBlock &b = net->blocks[i];
    TRY(bind_block(b, sh, levels, arena + off, x));
    b.g_out = nullptr, b.g_x = nullptr, b.g_tmp = nullptr;
    if (g_stage_hook) g_stage_hook(i, 0);
    TRY(mink_block_forward(&b, ex));
    x = b.out;
    off += sh.act + sh.stats;
  }
  net->out = const_cast<float *>(x);
  net->out_rows = levels[level].n;
  return MINK_OK;
}

int mink_net_backward(MinkNet *net, const MinkLevelMaps *levels, int32_t n_levels, float *arena, int64_t arena_floats,
                      const float *g_out, float *grad_arena, int64_t grad_floats, const MinkExec *ex, void *const *done_events) {
  MINK_REQUIRE(net && levels && arena && g_out && grad_arena && ex && net->blocks && net->n_blocks >= 1, "net_backward: bad arguments");
  TRY(ensure_events());
  // the layout of the forward pass, recomputed (another forward pass of the same net may have re-bound the descriptors)
  std::vector<BlockShape> shapes(net->n_blocks);
  std::vector<int64_t> offs(net->n_blocks);
  int level = 0;
  int64_t off = 0, goff_total = 0;
  for (int i = 0; i < net->n_blocks; ++i) {
    TRY(block_shape(net, levels, n_levels, i, level, shapes[i]));
    offs[i] = off;
    off += shapes[i].act + shapes[i].stats;
    goff_total += shapes[i].gtmp + shapes[i].gx;
  }
  MINK_REQUIRE(off <= arena_floats && goff_total <= grad_floats, "net_backward: arena too small (mink_net_sizes)");
  hipStream_t st = (hipStream_t)ex->compute, wst = (hipStream_t)ex->wgrad;
  const float *g = g_out;
  int64_t goff = 0;
  PendingGrad pending;
  for (int i = net->n_blocks - 1; i >= 0; --i) {
    const BlockShape &sh = shapes[i];
    MinkBasicBlock &b = net->blocks[i];
    const float *x = i == 0 ? net->stem.out : arena + offs[i - 1] + 3 * shapes[i - 1].n_out * net->blocks[i - 1].conv1.cout;
    TRY(bind_block(b, sh, levels, arena + offs[i], x));
    b.g_out = g, b.g_tmp = grad_arena + goff, b.g_x = grad_arena + goff + sh.gtmp;
    if (g_stage_hook) g_stage_hook(i, 1);
    // (an identity block leaves its input gradient to the block before it, which is on the same level with the same width)
    const bool may_defer = i > 0 && shapes[i - 1].level_out == sh.level_out && net->blocks[i - 1].conv1.cout == b.conv1.cin;
    PendingGrad next;
    TRY(block_backward_impl(&b, ex, &pending, may_defer ? &next : nullptr));
    pending = next;
    const bool want_event = done_events && done_events[i];
    if (want_event || g_block_done_hook) {
      // everything this block wrote into parameter-gradient buffers is complete behind this point of `wgrad`: the weight
      // gradients run there, the batch-norm gradients on `compute` / `branch`
      if (wst != st) TRY(order_after(wst, st, 7));
      if ((hipStream_t)ex->branch != st && (hipStream_t)ex->branch != wst) TRY(order_after(wst, (hipStream_t)ex->branch, 4));
      if (want_event) MINK_HIP(hipEventRecord((hipEvent_t)done_events[i], wst));
      if (g_block_done_hook) g_block_done_hook(i);
    }
    g = b.g_x;
    goff += sh.gtmp + sh.gx;
  }
  if (g_stage_hook) g_stage_hook(-1, 1);
  if (net->with_stem) {
    net->stem.g_out = g;
    TRY(mink_stem_backward(&net->stem, ex));
  }
  net->g_stem_out = g;
  return MINK_OK;
}

}  // extern "C"
