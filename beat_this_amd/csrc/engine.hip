// C ABI + forward orchestration of the BeatThis network on one MI355X.
// Follows BeatThis.forward (beat_this/model/beat_tracker.py:188-192): frontend (stem, three
// partial-transformer + conv blocks, linear), six RoFormer layers, final norm + SumHead.
// All launches go to the caller's stream; scratch comes from the caller's workspace.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/beat_this_amd.h"
#include "common.h"
#include "kernels.h"

// BT_PREC_F32X3 attention kernel of the forward (bt_attn_frag_args.x3): 4 = two query blocks per wave on the hand-scheduled
// key loop (round 4), 2 = the compiler-scheduled 64-key-tile kernel of round 3 (kept for A/B builds: -DBT_X3_ATTN=2)
#ifndef BT_X3_ATTN
#define BT_X3_ATTN 4
#endif

static thread_local std::string g_err;
static int bt_set_error(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

// Optional per-launch timing with HIP events on the caller's stream (bench.py roofline leg); state lives in the engine
// handle (reentrant per handle, like everything else).  Off by default: a normal bt_forward records nothing and never
// synchronises.
namespace prof {
struct Rec { int cat; hipEvent_t a, b; };
struct State { bool on = false; std::vector<Rec> recs; };
struct Scope {
  State* st; size_t idx; hipStream_t s;
  Scope(State* state, int cat, hipStream_t stream) : st(state && state->on ? state : nullptr), idx(0), s(stream) {
    if (!st) return;
    Rec r; r.cat = cat;
    (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, s);
    idx = st->recs.size(); st->recs.push_back(r);
  }
  ~Scope() { if (st) (void)hipEventRecord(st->recs[idx].b, s); }
};
}  // namespace prof

// a captured forward of bt_audio2beats_enqueue: key = everything the recorded launches depend on
struct FwdGraph { int B, T, prec; void* ws; hipGraphExec_t exec; unsigned long stamp; };

struct bt_engine {
  bt_model_desc d;
  prof::State prof;
  std::mutex mu;   // one engine may be shared by host threads (each on its own stream): the one-call path's bookkeeping -- graph cache, capture stream, options -- is serialised; the launches themselves go to the callers' streams
  std::vector<FwdGraph> graphs;   // (at most 8, least recently used goes first; dropped when an option changes)
  unsigned long graph_clock = 0;
  hipStream_t cap_stream = nullptr;   // private stream the forward is recorded on (the caller's may be the legacy default stream, which cannot capture)
  std::vector<int> warm;   // (precision << 8 | chunks) that ran plainly once: the kernels a shape selects are loaded / configured before they are recorded
  int x3_attn_p16 = 1;   // BT_OPT_X3_ATTN_P16 (default chosen by the flip-soak rule: DESIGN.md section 3)
  int x3_gemm_fp8 = 0;   // BT_OPT_X3_GEMM_FP8
};
enum { CAT_STEM = 0, CAT_QKV, CAT_ATTN_FLASH, CAT_OUT, CAT_FF1, CAT_FF2, CAT_CONV, CAT_LINEAR,
       CAT_HEAD, CAT_FF_FUSED, CAT_ATTN_FREQ_FUSED, CAT_LAYER_TAIL, CAT_COUNT };

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Workspace {
  float* xa; float* xb; float* xm; float* gates; void* xmb;
  void* qkv; void* ao; void* hid;
  void* qf; void* kf; void* vf; float* gates_h; int nbp;  // fragment-major attention operands (half path)
  float* ssq[2];  // [D / 64][B T] partial row sums of squares of the main residual stream (ping-pong)
  int* status;    // BT_PREC_F32X3 range flag: the FIRST word of the workspace (include/beat_this_amd.h)
  int* fix_mask;  // BT_PREC_F32X3: overflow map of the attention launches, [(sequences x heads)][nbp] words (attn2.hip)
  int x3_gemm_fp8;              // BT_OPT_X3_GEMM_FP8 of this forward
  int x3_attn, x3_attn_front;   // bt_attn_frag_args.x3 of this forward's attention launches, main layers / frontend (kernel choice + BT_X3_P16)
  size_t total;
};

Workspace carve(char* base, int B, int T, int D, int ff_mult, int prec) {
  const bool x3 = prec == BT_PREC_F32X3;
  if (x3) prec = BT_PREC_F32;   // (fp32 activations; hl32 planes = 4 bytes per element as well)
  const size_t es = prec == BT_PREC_F32 ? 4 : 2;
  const size_t bt = (size_t)B * T;
  const size_t dmax = std::max<size_t>(1024, D);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return base ? base + o : (char*)nullptr; };
  Workspace w;
  w.status = (int*)take(256);
  w.x3_attn = w.x3_attn_front = BT_X3_ATTN;
  w.x3_gemm_fp8 = 0;
  w.xa = (float*)take(bt * 1024 * 4);
  w.xb = (float*)take(bt * 1024 * 4);
  w.xm = (float*)take(bt * D * 4);
  w.xmb = take(bt * D * (x3 ? 4 : 2));  // half (x3: hl32) shadow of the main residual stream (A operand of the wide GEMMs)
  w.gates = (float*)take(bt * std::max(32, D / 32) * 4);
  w.qkv = take(bt * 3 * dmax * es);
  w.ao = take(bt * dmax * es);
  w.hid = take(bt * std::max<size_t>(4 * 1024, (size_t)ff_mult * D) * es);  // FF hidden activation / conv shadow
  w.nbp = attn_frag_blocks(T);
  w.qf = w.kf = w.vf = nullptr; w.gates_h = nullptr;
  if (prec == BT_PREC_HALF || x3) {  // (sequences x heads) = 32 B in the frontend, (D / 32) B in the main layers
    const size_t sh = (size_t)B * std::max(32, D / 32);
    const size_t blk = x3 ? 4096 : 2048;  // bytes per 32-token block (x3: [hi block | lo block])
    w.qf = take(sh * w.nbp * blk); w.kf = take(sh * w.nbp * blk); w.vf = take(sh * w.nbp * blk);
    w.gates_h = (float*)take(sh * w.nbp * 32 * 4);
    w.fix_mask = x3 ? (int*)take(sh * w.nbp * 4) : nullptr;
    w.ssq[0] = (float*)take(bt * (D / 64 + 1) * 4);
    w.ssq[1] = (float*)take(bt * (D / 64 + 1) * 4);
  } else {
    w.ssq[0] = w.ssq[1] = nullptr;
    w.fix_mask = nullptr;
  }
  w.total = off;
  return w;
}

#define CHECK_RC(what)                                                         \
  if (_rc != 0) {                                                              \
    char buf[160];                                                             \
    snprintf(buf, sizeof buf, "%s failed (code %d: %s)", what, _rc,            \
             _rc > 0 ? hipGetErrorString((hipError_t)_rc) : "bad arguments");  \
    return bt_set_error(_rc > 0 ? BT_ERR_HIP : BT_ERR_ARG, buf);               \
  }
#define LAUNCH(expr, what) \
  do { int _rc = (expr); CHECK_RC(what) } while (0)
#define LAUNCH_CAT(cat, st, expr, what) \
  do { int _rc; { prof::Scope _ps(pf, cat, st); _rc = (expr); } CHECK_RC(what) } while (0)

// mode 0: main transformer (sequences = chunks, tokens = frames)
// mode 1: frequency direction (sequences = (b,t), tokens = f)      -- attnff_fused_kernel only
// mode 2: time direction      (sequences = (b,f), tokens = t)      -- rows permuted around attn_flash
// Main transformer layer in BT_PREC_HALF on the gemm3 / attn2 kernels.  ws.ssq[0] holds the partial row sums of
// squares of x on entry and on exit (written by the producer of x: frontend.linear or the previous FF2), ws.ssq[1]
// those of x after the attention half.
int run_layer_half(prof::State* pf, const bt_pair_weights& pw, const float* rope, const Workspace& ws, int B, int T,
                   int ff_mult, hipStream_t s) {
  const int D = pw.dim, H = pw.heads, HID = ff_mult * D;
  const int M = B * T;
  const int parts = D / 64;
  Gemm3P g;
  memset(&g, 0, sizeof g);
  g.A = ws.xmb; g.lda = D; g.M = M; g.K = D; g.W = pw.w_qkvg[BT_PREC_HALF]; g.N = 3 * D + H; g.epi = G3_QKV;
  g.ssq_in = ws.ssq[0]; g.ssq_parts = parts;
  g.n_seq = B; g.L = T; g.nblk = (T + 31) / 32; g.nbp = ws.nbp; g.heads = H; g.inner = D; g.rope = rope;
  g.qf = ws.qf; g.kf = ws.kf; g.vf = ws.vf; g.gates = ws.gates_h; g.b_gates = pw.b_gates;
  LAUNCH_CAT(CAT_QKV, s, launch_gemm3(g, s), "qkv gemm");
  AttnFragP a;
  memset(&a, 0, sizeof a);
  a.q = ws.qf; a.k = ws.kf; a.v = ws.vf; a.gates = ws.gates_h; a.out = ws.ao; a.n_seq = B; a.L = T; a.heads = H;
  a.inner = D; a.nbp = ws.nbp; a.o_div = 1; a.o_outer = T; a.o_inner = 0; a.o_tok = 1;
  LAUNCH_CAT(CAT_ATTN_FLASH, s, launch_attn_frag(a, s), "attention");
  if (pw.w_tail_frag && layer_tail_supported(D, HID)) {
    // out-projection + FF1 + FF2 in ONE launch: x, its half shadow and the statistics of the new x written once
    LayerTailP t;
    t.x = ws.xm; t.M = M; t.C = D; t.hidden = HID; t.ao = ws.ao; t.wfrag = pw.w_tail_frag; t.b1 = pw.b_ff1; t.b2 = pw.b_ff2;
    t.xb = ws.xmb; t.ssq_out = ws.ssq[0];
    LAUNCH_CAT(CAT_LAYER_TAIL, s, launch_layer_tail(t, s), "layer tail (out-projection + feed-forward)");
    return BT_OK;
  }
  memset(&g, 0, sizeof g);
  g.A = ws.ao; g.lda = D; g.M = M; g.K = D; g.W = pw.w_out[BT_PREC_HALF]; g.N = D; g.epi = G3_RESID;
  g.x = ws.xm; g.ldx = D; g.xb = ws.xmb; g.ssq_out = ws.ssq[1];
  LAUNCH_CAT(CAT_OUT, s, launch_gemm3(g, s), "out-proj gemm");
  memset(&g, 0, sizeof g);
  g.A = ws.xmb; g.lda = D; g.M = M; g.K = D; g.W = pw.w_ff1[BT_PREC_HALF]; g.N = HID; g.epi = G3_FF1;
  g.bias = pw.b_ff1; g.ssq_in = ws.ssq[1]; g.ssq_parts = parts; g.out = ws.hid; g.ldo = HID;
  LAUNCH_CAT(CAT_FF1, s, launch_gemm3(g, s), "ff1 gemm");
  memset(&g, 0, sizeof g);
  g.A = ws.hid; g.lda = HID; g.M = M; g.K = HID; g.W = pw.w_ff2[BT_PREC_HALF]; g.N = D; g.epi = G3_RESID;
  g.bias = pw.b_ff2; g.x = ws.xm; g.ldx = D; g.xb = ws.xmb; g.ssq_out = ws.ssq[0];
  LAUNCH_CAT(CAT_FF2, s, launch_gemm3(g, s), "ff2 gemm");
  return BT_OK;
}

// Main transformer layer in BT_PREC_F32X3 on the same kernels with hi + lo operands (gemm3.hip X3, attn2.hip
// attn_frag_x3_kernel): the fp32 residual stream ws.xm is shadowed by hl32 planes (ws.xmb), the attention output and
// the hidden activation travel as hl32 planes, q / k / v as 4 KB [hi | lo] fragment blocks; statistics as in the half layer.
// xmb_f8: the shadow of the residual stream this layer finds is hl8 (BT_OPT_X3_GEMM_FP8 = 2: written by the previous layer's FF2 or
// by frontend.linear); next_f8: this layer's FF2 leaves it in that form for the next one
int run_layer_x3(prof::State* pf, const bt_pair_weights& pw, const float* rope, const Workspace& ws, int B, int T,
                 int ff_mult, hipStream_t s, bool xmb_f8 = false, bool next_f8 = false) {
  const int D = pw.dim, H = pw.heads, HID = ff_mult * D;
  const int M = B * T;
  const int parts = D / 64;
  Gemm3P g;
  memset(&g, 0, sizeof g);
  g.A = ws.xmb; g.lda = D; g.M = M; g.K = D; g.W = xmb_f8 ? pw.w_qkvg_f8 : pw.w_qkvg_x3; g.N = 3 * D + H; g.epi = G3_QKV;
  g.ssq_in = ws.ssq[0]; g.ssq_parts = parts; g.x3 = 1 | (xmb_f8 ? G3_X3_F8 : 0); g.status = ws.status;
  g.n_seq = B; g.L = T; g.nblk = (T + 31) / 32; g.nbp = ws.nbp; g.heads = H; g.inner = D; g.rope = rope;
  g.qf = ws.qf; g.kf = ws.kf; g.vf = ws.vf; g.gates = ws.gates_h; g.b_gates = pw.b_gates;
  LAUNCH_CAT(CAT_QKV, s, launch_gemm3(g, s), "qkv gemm (hi + lo)");
  AttnFragP a;
  memset(&a, 0, sizeof a);
  a.q = ws.qf; a.k = ws.kf; a.v = ws.vf; a.gates = ws.gates_h; a.out = ws.ao; a.n_seq = B; a.L = T; a.heads = H;
  a.inner = D; a.nbp = ws.nbp; a.o_div = 1; a.o_outer = T; a.o_inner = 0; a.o_tok = 1;
  // BT_OPT_X3_GEMM_FP8 = 2: out-projection (and QKV) on hl8 operands as well -- the attention writes its rows in that form
  const bool out8 = ws.x3_gemm_fp8 >= 2 && pw.w_out_f8;
  a.x3 = ws.x3_attn; a.out_f32 = out8 ? 2 : 0; a.status = ws.status; a.fix_mask = ws.fix_mask;   // (which x3 attention kernel / arithmetic: attn2.hip launch_attn_frag)
  LAUNCH_CAT(CAT_ATTN_FLASH, s, launch_attn_frag(a, s), "attention (hi + lo)");
  // BT_OPT_X3_GEMM_FP8 >= 1: the feed-forward GEMMs on hl8 operands (fp8 cross terms): the out-projection leaves its shadow of
  // the residual stream in that form, FF1 its hidden activation; FF2's shadow feeds the next layer's QKV and stays hl32
  const bool ff8 = ws.x3_gemm_fp8 >= 1 && pw.w_ff1_f8 && pw.w_ff2_f8;
  memset(&g, 0, sizeof g);
  g.A = ws.ao; g.lda = D; g.M = M; g.K = D; g.W = out8 ? pw.w_out_f8 : pw.w_out_x3; g.N = D; g.epi = G3_RESID; g.status = ws.status;
  g.x3 = 1 | (out8 ? G3_X3_F8 : 0) | (ff8 ? G3_X3_OUT_F8 : 0);
  g.x = ws.xm; g.ldx = D; g.xb = ws.xmb; g.ssq_out = ws.ssq[1];
  LAUNCH_CAT(CAT_OUT, s, launch_gemm3(g, s), "out-proj gemm (hi + lo)");
  memset(&g, 0, sizeof g);
  g.A = ws.xmb; g.lda = D; g.M = M; g.K = D; g.W = ff8 ? pw.w_ff1_f8 : pw.w_ff1_x3; g.N = HID; g.epi = G3_FF1; g.status = ws.status;
  g.x3 = 1 | (ff8 ? G3_X3_F8 | G3_X3_OUT_F8 : 0);
  g.bias = pw.b_ff1; g.ssq_in = ws.ssq[1]; g.ssq_parts = parts; g.out = ws.hid; g.ldo = HID;
  LAUNCH_CAT(CAT_FF1, s, launch_gemm3(g, s), "ff1 gemm (hi + lo)");
  memset(&g, 0, sizeof g);
  g.A = ws.hid; g.lda = HID; g.M = M; g.K = HID; g.W = ff8 ? pw.w_ff2_f8 : pw.w_ff2_x3; g.N = D; g.epi = G3_RESID; g.status = ws.status;
  g.x3 = 1 | (ff8 ? G3_X3_F8 : 0) | (next_f8 ? G3_X3_OUT_F8 : 0);
  g.bias = pw.b_ff2; g.x = ws.xm; g.ldx = D; g.xb = ws.xmb; g.ssq_out = ws.ssq[0];
  LAUNCH_CAT(CAT_FF2, s, launch_gemm3(g, s), "ff2 gemm (hi + lo)");
  return BT_OK;
}

// half shadow of the residual stream written by the fused out-projection + FF kernel (time-direction half; A operand of
// the following frontend conv on gemm3)
inline bool pair_fused2_ok(const bt_pair_weights& pw, int prec) {
  return pw.dim <= 128 && pw.w_ff_frag[prec] && pw.w_outff_frag[prec] && pw.w_attnff_frag[prec];
}

// part (mode 0 only; bt_forward_unit): 0 = the whole pair, 1 = the attention half (x += Attention(x)) only, 2 = the
// feed-forward half (x += FeedForward(x)) only
int run_pair(prof::State* pf, const bt_pair_weights& pw, const float* rope, float* x, void* xshadow, const Workspace& ws,
             int B, int T, int F, int mode, int prec, hipStream_t s, void* out_shadow = nullptr, int ff_mult = 4,
             bool x3 = false, int part = 0) {
  const int C = pw.dim, H = pw.heads, HID = ff_mult * C;
  // BT_PREC_F32X3: prec is BT_PREC_F32 for everything but the plain GEMMs, which take the [hi | lo] half weights
  const int gp = x3 ? BT_PREC_F32X3 : prec, wp = x3 ? BT_PREC_HALF : prec;
  const long M = (long)B * T * F;
  if (M > 0x7fffffffL) return bt_set_error(BT_ERR_ARG, "batch too large for one forward call");
  GemmP g;
  // main layers in half mode read the half shadow of x (half the operand bytes, no conversion in the k-loop)
  const bool shadow = xshadow != nullptr && prec == BT_PREC_HALF;
  const bool fused_ok = C <= 128 && pw.w_ff_frag[prec];
  const bool fused2_ok = fused_ok && pw.w_outff_frag[prec] && pw.w_attnff_frag[prec];
  // BT_PREC_F32X3: the register-chained halves on (hi, lo) operands too when their split streams were packed
  const bool f2x3 = x3 && pw.w_outff_frag_x3 && pw.w_attnff_frag_x3;
  auto outff = [&]() -> int {  // x += to_out(ws.ao); x += FF(x) in one launch
    FusedOutFFP f;
    f.x = x; f.M = M; f.C = C; f.ao = ws.ao; f.wfrag = f2x3 ? pw.w_outff_frag_x3 : pw.w_outff_frag[prec];
    f.b1 = pw.b_ff1; f.b2 = pw.b_ff2; f.xb = out_shadow; f.abl = 0;
    LAUNCH_CAT(CAT_FF_FUSED, s, launch_outff_fused(f, f2x3 ? BT_PREC_F32X3 : prec, s), "fused out-projection + feed-forward");
    return BT_OK;
  };
  if (mode == 1 && fused2_ok) {  // whole frequency-direction half (attention + FF) in one register-resident kernel
    FusedAttnFFP f;
    f.x = x; f.M = M; f.C = C; f.b_gates = pw.b_gates; f.rope = rope;
    f.wfrag = f2x3 ? pw.w_attnff_frag_x3 : pw.w_attnff_frag[prec]; f.b1 = pw.b_ff1; f.b2 = pw.b_ff2;
    LAUNCH_CAT(CAT_ATTN_FREQ_FUSED, s, launch_attnff_fused(f, f2x3 ? BT_PREC_F32X3 : prec, s),
               "fused frequency attention + feed-forward");
    return BT_OK;
  }
  if (mode == 1)  // (pack.py always supplies the fused-half weight streams for the frontend's pairs)
    return bt_set_error(BT_ERR_ARG, "frequency-direction half needs bt_pair_weights.w_attnff_frag");
  // BT_PREC_F32X3 time direction on the fragment-major kernels: hi + lo QKV blocks straight from the projection, the
  // attention's fp32 output feeds the (hi, lo) out-projection + FF kernel
  const bool t2x3 = mode == 2 && f2x3 && fused2_ok && pw.w_qkv_frag_x3 && ws.qf;
  if (part == 2) {
    // (feed-forward half only: nothing of the attention half runs)
  } else if (mode == 2 && fused_ok && ((prec == BT_PREC_HALF && pw.w_qkv_frag) || t2x3)) {
    // time direction: fragment-major QKV straight from the projection, flash attention on it
    QkvFrontP qp;
    memset(&qp, 0, sizeof qp);
    qp.x = x; qp.B = B; qp.T = T; qp.F = F; qp.C = C; qp.wfrag = t2x3 ? pw.w_qkv_frag_x3 : pw.w_qkv_frag;
    qp.b_gates = pw.b_gates; qp.rope = rope;
    qp.q = ws.qf; qp.k = ws.kf; qp.v = ws.vf; qp.gates = ws.gates_h; qp.nbp = ws.nbp;
    qp.x3 = t2x3; qp.status = ws.status;
    LAUNCH_CAT(CAT_QKV, s, launch_qkv_front(qp, s), "frontend qkv projection");
    AttnFragP a;
    memset(&a, 0, sizeof a);
    a.q = ws.qf; a.k = ws.kf; a.v = ws.vf; a.gates = ws.gates_h; a.out = ws.ao; a.n_seq = B * F; a.L = T; a.heads = H;
    a.inner = C; a.nbp = ws.nbp; a.o_div = F; a.o_outer = (long)T * F; a.o_inner = 1; a.o_tok = F;
    a.x3 = t2x3 ? ws.x3_attn_front : 0; a.out_f32 = 1; a.status = ws.status; a.fix_mask = ws.fix_mask;
    LAUNCH_CAT(CAT_ATTN_FLASH, s, launch_attn_frag(a, s), "time attention");
    if (fused2_ok) return outff();
    memset(&g, 0, sizeof g);
    g.A = ws.ao; g.lda = C; g.W = pw.w_out[wp]; g.M = (int)M; g.N = C; g.K = C;
    g.epi = GEMM_EPI_RESID; g.flags = 0; g.x = x; g.ldx = C; g.xb = nullptr;
    LAUNCH_CAT(CAT_OUT, s, launch_gemm(g, gp, s), "out-proj gemm");
  } else {
  // ---- q|k|v|gates = RMSNorm(x) . W^T, RoPE, sigmoid ------------------------------------
  memset(&g, 0, sizeof g);
  g.A = shadow ? xshadow : (const void*)x; g.lda = C; g.W = pw.w_qkvg[wp]; g.M = (int)M; g.N = 3 * C + H; g.K = C;
  g.epi = GEMM_EPI_QKV; g.flags = GEMM_F_RMS | (shadow ? 0 : GEMM_F_A_F32) | (mode == 2 ? GEMM_F_ROWMAP : 0);
  g.bias = pw.b_gates; g.out = ws.qkv; g.ldo = 3 * C; g.gates = ws.gates; g.inner = C; g.heads = H;
  g.rope = rope; g.map_T = T; g.map_F = F;
  if (mode == 0) { g.pdiv = 1; g.pmod = T; }
  else if (mode == 1) { g.pdiv = 1; g.pmod = F; }
  else { g.pdiv = F; g.pmod = T; }
  LAUNCH_CAT(CAT_QKV, s, launch_gemm(g, gp, s), "qkv gemm");
  // ---- attention ------------------------------------------------------------------------
  AttnP a;
  memset(&a, 0, sizeof a);
  a.qkv = ws.qkv; a.ld = 3 * C; a.gates = ws.gates; a.out = ws.ao; a.heads = H; a.inner = C;
  if (mode == 2) {
    a.n_seq = B * F; a.L = T; a.o_div = F; a.o_outer = (long)T * F; a.o_inner = 1; a.o_tok = F;
    LAUNCH_CAT(CAT_ATTN_FLASH, s, launch_attn_flash(a, gp, s), "time attention");
  } else {
    a.n_seq = B; a.L = T; a.o_div = 1; a.o_outer = T; a.o_inner = 0; a.o_tok = 1;
    LAUNCH_CAT(CAT_ATTN_FLASH, s, launch_attn_flash(a, gp, s), "attention");
  }
  if (mode == 2 && fused2_ok) return outff();
  // ---- x += ao . Wout^T -------------------------------------------------------------------
  memset(&g, 0, sizeof g);
  g.A = ws.ao; g.lda = C; g.W = pw.w_out[wp]; g.M = (int)M; g.N = C; g.K = C;
  g.epi = GEMM_EPI_RESID; g.flags = 0; g.x = x; g.ldx = C; g.xb = shadow ? xshadow : nullptr;
  LAUNCH_CAT(CAT_OUT, s, launch_gemm(g, gp, s), "out-proj gemm");
  }
  if (part == 1) return BT_OK;
  if (fused_ok) {
    FusedFFP ff;
    ff.x = x; ff.M = M; ff.C = C; ff.wfrag = pw.w_ff_frag[prec]; ff.b1 = pw.b_ff1; ff.b2 = pw.b_ff2;
    ff.xb = shadow ? xshadow : nullptr;
    LAUNCH_CAT(CAT_FF_FUSED, s, launch_ff_fused(ff, prec, s), "fused feed-forward");
    return BT_OK;
  }
  // ---- h = gelu(RMSNorm(x) . W1^T + b1) ------------------------------------------------------
  memset(&g, 0, sizeof g);
  g.A = shadow ? xshadow : (const void*)x; g.lda = C; g.W = pw.w_ff1[wp]; g.M = (int)M; g.N = HID; g.K = C;
  g.epi = GEMM_EPI_STORE; g.flags = GEMM_F_RMS | (shadow ? 0 : GEMM_F_A_F32) | GEMM_F_BIAS | GEMM_F_GELU;
  g.bias = pw.b_ff1; g.out = ws.hid; g.ldo = HID;
  LAUNCH_CAT(CAT_FF1, s, launch_gemm(g, gp, s), "ff1 gemm");
  // ---- x += h . W2^T + b2 ---------------------------------------------------------------------
  memset(&g, 0, sizeof g);
  g.A = ws.hid; g.lda = HID; g.W = pw.w_ff2[wp]; g.M = (int)M; g.N = C; g.K = HID;
  g.epi = GEMM_EPI_RESID; g.flags = GEMM_F_BIAS; g.bias = pw.b_ff2; g.x = x; g.ldx = C; g.xb = shadow ? xshadow : nullptr;
  LAUNCH_CAT(CAT_FF2, s, launch_gemm(g, gp, s), "ff2 gemm");
  return BT_OK;
}

}  // namespace

extern "C" {

const char* bt_last_error(void) { return g_err.c_str(); }
int bt_version(void) { return BT_ABI_VERSION; }
int bt_half_is_bf16(void) { return BT_HALF_IS_BF16; }
void bt_struct_sizes(int32_t* out) {
  out[0] = (int32_t)sizeof(bt_pair_weights); out[1] = (int32_t)sizeof(bt_model_desc);
  out[2] = (int32_t)sizeof(bt_logmel_tables); out[3] = (int32_t)sizeof(bt_gemm_args);
  out[4] = (int32_t)sizeof(bt_attn_args); out[5] = (int32_t)offsetof(bt_model_desc, layers);
  out[6] = (int32_t)offsetof(bt_model_desc, rope);
  out[7] = (int32_t)sizeof(bt_attn_frag_args);
  out[8] = (int32_t)sizeof(bt_gemm3_args);
}

int bt_engine_create(const bt_model_desc* desc, bt_engine** out) {
  if (!desc || !out) return bt_set_error(BT_ERR_ARG, "null argument");
  if (desc->transformer_dim % 32 || desc->transformer_dim < 32 || desc->transformer_dim > 1024 ||
      desc->n_layers < 0 || desc->n_layers > BT_MAX_LAYERS || desc->ff_mult < 1 || desc->ff_mult > 16)
    return bt_set_error(BT_ERR_ARG, "unsupported transformer_dim / n_layers / ff_mult");
  bt_engine* e = new bt_engine;
  e->d = *desc;
  *out = e;
  return BT_OK;
}
static void drop_graphs(bt_engine* e) {
  for (auto& g : e->graphs) (void)hipGraphExecDestroy(g.exec);
  e->graphs.clear();
  e->warm.clear();   // (another arithmetic option selects other kernels: they run plainly once before they are recorded again)
}
void bt_engine_destroy(bt_engine* e) {
  if (!e) return;
  for (auto& r : e->prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  drop_graphs(e);
  if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
  delete e;
}

int bt_engine_set_option(bt_engine* e, int option, int value) {
  if (!e) return bt_set_error(BT_ERR_ARG, "null argument");
  // (a captured forward replays the kernels it was recorded with: an option change drops them)
  std::lock_guard<std::mutex> lock(e->mu);
  if (option == BT_OPT_X3_ATTN_P16 && value >= 0 && value <= 3) { if (e->x3_attn_p16 != value) drop_graphs(e); e->x3_attn_p16 = value; return BT_OK; }
  if (option == BT_OPT_X3_GEMM_FP8 && value >= 0 && value <= 2) { if (e->x3_gemm_fp8 != value) drop_graphs(e); e->x3_gemm_fp8 = value; return BT_OK; }
  return bt_set_error(BT_ERR_ARG, "unknown engine option / value");
}
int bt_engine_get_option(const bt_engine* e, int option, int* value) {
  if (!e || !value) return bt_set_error(BT_ERR_ARG, "null argument");
  if (option == BT_OPT_X3_ATTN_P16) { *value = e->x3_attn_p16; return BT_OK; }
  if (option == BT_OPT_X3_GEMM_FP8) { *value = e->x3_gemm_fp8; return BT_OK; }
  return bt_set_error(BT_ERR_ARG, "unknown engine option");
}

size_t bt_workspace_bytes(const bt_engine* e, int B, int T, int prec) {
  if (!e || B <= 0 || T <= 0) return 0;
  return carve(nullptr, B, T, e->d.transformer_dim, e->d.ff_mult, prec).total;
}

int bt_forward(bt_engine* e, void* stream, int prec, const float* d_spect, int B, int T, void* d_ws, size_t ws_bytes,
               float* d_beat, float* d_downbeat) {
  return bt_forward_stages(e, stream, prec, 0, 2, d_spect, B, T, d_ws, ws_bytes, nullptr, d_beat, d_downbeat);
}

int bt_forward_stages(bt_engine* e, void* stream, int prec, int first, int last, const float* d_spect, int B, int T, void* d_ws,
                      size_t ws_bytes, float* d_out, float* d_beat, float* d_downbeat) {
  if (!e || !d_spect || !d_ws) return bt_set_error(BT_ERR_ARG, "null argument");
  if (first < 0 || last > 2 || first > last) return bt_set_error(BT_ERR_ARG, "stages: need 0 <= first <= last <= 2");
  if (last == 2 ? (!d_beat || !d_downbeat) : !d_out) return bt_set_error(BT_ERR_ARG, "null output");
  const int max_T = e->d.rope_len > 0 ? e->d.rope_len : 1536;   // rows of the rotary table
  if (B <= 0 || T <= 0 || T > max_T)
    return bt_set_error(BT_ERR_ARG, "need B >= 1 and 1 <= T <= " + std::to_string(max_T) + " (bt_model_desc.rope_len: rows of the rotary table)");
  if (prec != BT_PREC_F32 && prec != BT_PREC_HALF && prec != BT_PREC_F32X3)
    return bt_set_error(BT_ERR_ARG, "unknown precision");
  const bt_model_desc& d = e->d;
  prof::State* pf = &e->prof;
  const int D = d.transformer_dim;
  Workspace ws = carve((char*)d_ws, B, T, D, d.ff_mult, prec);
  if (ws.total > ws_bytes) return bt_set_error(BT_ERR_WORKSPACE, "workspace too small");
  ws.x3_attn = BT_X3_ATTN | ((e->x3_attn_p16 == 1 || e->x3_attn_p16 == 2) ? BT_X3_P16 : 0);   // (3: the frontend only -- a soak variant)
  ws.x3_attn_front = BT_X3_ATTN | (e->x3_attn_p16 >= 2 ? BT_X3_P16 : 0);
  ws.x3_gemm_fp8 = e->x3_gemm_fp8;
  hipStream_t s = (hipStream_t)stream;
  const bool x3 = prec == BT_PREC_F32X3;
  if (x3) {
    if (BT_HALF_IS_BF16) return bt_set_error(BT_ERR_ARG, "BT_PREC_F32X3 needs an IEEE fp16 build");
    prec = BT_PREC_F32;
    // range flag of this forward (first word of the workspace): cleared here; bit 0 is ORed by the gemm3 / attention / QKV
    // kernels when a value beyond the fp16 range goes through a split, bit 1 by whatever ends the call (head, final norm,
    // stage exit) when its output is not finite -- which is where an overflow in any other splitting kernel ends up
    LAUNCH(launch_clear_words(ws.status, 1, s), "clearing the range flag");
  }
  const int gp = x3 ? BT_PREC_F32X3 : prec, wp = x3 ? BT_PREC_HALF : prec;   // plain GEMMs: launch / weight precision

  // the half shadow of the main residual stream is maintained by the gemm2 / gemm3 epilogues only
  const bool use_shadow = prec == BT_PREC_HALF && D >= 128 && D % 64 == 0;
  // main layers on gemm3 + fragment-major attention (needs q | k | v column blocks that are whole 128-tiles)
  const bool fast_layers = use_shadow && D % 128 == 0 && (long)B * T * d.ff_mult * D * 2 < 0x7fffffffL;
  // BT_PREC_F32X3 on the same kernels (hl32 operands): needs the hl32 weights of every layer
  bool fast_x3 = x3 && D >= 128 && D % 128 == 0 && (long)B * T * d.ff_mult * D * 4 < 0x7fffffffL;
  for (int l = 0; fast_x3 && l < d.n_layers; ++l)
    fast_x3 = d.layers[l].w_qkvg_x3 && d.layers[l].w_out_x3 && d.layers[l].w_ff1_x3 && d.layers[l].w_ff2_x3;

  // frontend.linear on gemm3 (half A written by the last conv block) when its shape fits
  bool lin3 = false;
  if (fast_layers || (fast_x3 && d.lin_w_x3)) {
    Gemm3P g;
    memset(&g, 0, sizeof g);
    g.lda = 1024; g.M = B * T; g.K = 1024; g.N = D; g.epi = G3_RESID; g.ldx = D; g.x = ws.xm; g.x3 = fast_x3;
    lin3 = gemm3_supported(g);
  }

  const size_t xm_bytes = (size_t)B * T * D * 4;
  bool xmb_f8 = false;   // the shadow of the main residual stream is hl8 (BT_OPT_X3_GEMM_FP8 = 2) when the first layer starts
  if (first == 2) {  // task_heads on a normalised [B,T,D] input
    if (!d.head_w_raw) return bt_set_error(BT_ERR_ARG, "stage entry at task_heads needs head_w_raw");
    HeadP hp;
    hp.x = d_spect; hp.w = d.head_w_raw; hp.b0 = d.head_b[0]; hp.b1 = d.head_b[1];
    hp.beat = d_beat; hp.downbeat = d_downbeat; hp.M = B * T; hp.D = D; hp.sum_head = d.sum_head; hp.prenorm = 1;
    hp.status = nullptr;
    LAUNCH_CAT(CAT_HEAD, s, launch_head(hp, s), "head");
    return BT_OK;
  }
  if (first == 1) {  // transformer_blocks on a [B,T,D] input: the residual stream and what its producer would have left
    if (hipMemcpyAsync(ws.xm, d_spect, xm_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return bt_set_error(BT_ERR_HIP, "copy of the stage input");
    if (use_shadow) LAUNCH_CAT(CAT_LINEAR, s, launch_shadow_ssq(ws.xm, ws.xmb, fast_layers ? ws.ssq[0] : nullptr, (long)B * T, D, s), "stage entry");
    if (fast_x3) LAUNCH_CAT(CAT_LINEAR, s, launch_shadow_ssq(ws.xm, ws.xmb, ws.ssq[0], (long)B * T, D, s, 1), "stage entry");
  }
  if (first == 0) {
  StemP sp;
  sp.spect = d_spect; sp.x = ws.xa; sp.bn1_scale = d.bn1_scale; sp.bn1_shift = d.bn1_shift;
  sp.w = d.stem_w; sp.bias = d.stem_b; sp.B = B; sp.T = T;
  LAUNCH_CAT(CAT_STEM, s, launch_stem(sp, s), "stem");

  float* x = ws.xa;
  float* xn = ws.xb;
  for (int blk = 0; blk < 3; ++blk) {
    const int C = 32 << blk, F = 32 >> blk;
    // conv of this block on gemm3 (LDS-DMA ring on the half shadow of x that the time-direction half leaves in ws.hid)
    Gemm3P cg;
    memset(&cg, 0, sizeof cg);
    cg.A = ws.hid; cg.lda = 2 * C; cg.M = B * T * (F / 2); cg.K = 6 * C; cg.N = 2 * C;
    cg.W = fast_x3 ? d.conv_w_x3[blk] : d.conv_w[blk][BT_PREC_HALF];
    cg.epi = G3_RESID; cg.no_resid = 1; cg.gelu = 1; cg.bias = d.conv_b[blk]; cg.ldx = 2 * C;
    cg.conv_C2 = 2 * C; cg.conv_T = T; cg.conv_F = F / 2; cg.x3 = fast_x3; cg.status = ws.status;
    const bool to_bf16 = blk == 2 && lin3;  // the last block's output is read by frontend.linear (gemm3) only
    cg.x = to_bf16 ? nullptr : xn; cg.xb = to_bf16 ? (void*)xn : nullptr;
    // (the first conv, N = 64, works as well but only pays 4 us for the 12 us its shadow write costs: it stays on gemm.hip)
    // x3: the shadow is the hl32 form written by the (hi, lo) out-projection + FF kernel of the time-direction half
    const bt_pair_weights& tw = d.front[blk][1];
    const bool conv3 = d.partial_transformers && cg.N >= 128 && cg.W &&
                       (fast_x3 ? (pair_fused2_ok(tw, prec) && tw.w_outff_frag_x3 && tw.w_attnff_frag_x3)
                                : (fast_layers && pair_fused2_ok(tw, prec))) &&
                       gemm3_supported(cg);
    // (x3: frontend.linear on gemm3 reads hl32 planes, which only the gemm3 form of the last convolution writes)
    if (blk == 2 && fast_x3 && !conv3) lin3 = false;
    if (d.partial_transformers) {
      int rc = run_pair(pf, d.front[blk][0], d.rope, x, nullptr, ws, B, T, F, 1, prec, s, nullptr, 4, x3);
      if (rc) return rc;
      rc = run_pair(pf, d.front[blk][1], d.rope, x, nullptr, ws, B, T, F, 2, prec, s, conv3 ? ws.hid : nullptr, 4, x3);
      if (rc) return rc;
    }
    if (conv3) {
      LAUNCH_CAT(CAT_CONV, s, launch_gemm3(cg, s), "frontend conv gemm");
      std::swap(x, xn);
      continue;
    }
    GemmP g;
    memset(&g, 0, sizeof g);
    g.A = x; g.W = d.conv_w[blk][wp]; g.M = B * T * (F / 2); g.N = 2 * C; g.K = 6 * C;
    g.epi = GEMM_EPI_STORE; g.flags = GEMM_F_CONV | GEMM_F_A_F32 | GEMM_F_BIAS | GEMM_F_GELU | GEMM_F_OUT_F32;
    // the last block's output is read by frontend.linear only: half when that runs on gemm3 (same rounding point as
    // the fp32 -> half conversion of its A operand, half the bytes both ways)
    if (blk == 2 && lin3) g.flags &= ~GEMM_F_OUT_F32;
    g.bias = d.conv_b[blk]; g.out = xn; g.ldo = 2 * C;
    g.conv_C2 = 2 * C; g.conv_T = T; g.conv_F = F / 2;
    LAUNCH_CAT(CAT_CONV, s, launch_gemm(g, gp, s), "frontend conv gemm");
    std::swap(x, xn);
  }
  if (lin3) {
    Gemm3P g;
    memset(&g, 0, sizeof g);
    g.A = x; g.lda = 1024; g.M = B * T; g.K = 1024; g.W = fast_x3 ? d.lin_w_x3 : d.lin_w[BT_PREC_HALF]; g.N = D; g.epi = G3_RESID;
    g.no_resid = 1; g.bias = d.lin_b; g.x = ws.xm; g.ldx = D; g.xb = ws.xmb; g.ssq_out = ws.ssq[0];
    xmb_f8 = fast_x3 && ws.x3_gemm_fp8 >= 2 && d.n_layers > 0 && d.layers[0].w_qkvg_f8;   // (layer 0's QKV reads it in that form)
    g.x3 = fast_x3 ? 1 | (xmb_f8 ? G3_X3_OUT_F8 : 0) : 0; g.status = ws.status;
    LAUNCH_CAT(CAT_LINEAR, s, launch_gemm3(g, s), "frontend linear gemm");
  } else {
    GemmP g;
    memset(&g, 0, sizeof g);
    g.A = x; g.lda = 1024; g.W = d.lin_w[wp]; g.M = B * T; g.N = D; g.K = 1024;
    g.epi = GEMM_EPI_STORE; g.flags = GEMM_F_A_F32 | GEMM_F_BIAS | GEMM_F_OUT_F32;
    g.bias = d.lin_b; g.out = ws.xm; g.ldo = D; g.xb = use_shadow ? ws.xmb : nullptr;
    g.ssq_out = fast_layers ? ws.ssq[0] : nullptr;
    LAUNCH_CAT(CAT_LINEAR, s, launch_gemm(g, gp, s), "frontend linear gemm");
    if (fast_x3) LAUNCH_CAT(CAT_LINEAR, s, launch_shadow_ssq(ws.xm, ws.xmb, ws.ssq[0], (long)B * T, D, s, 1), "hl32 shadow of the residual stream");
  }
  }  // first == 0
  if (last == 0) {
    if (hipMemcpyAsync(d_out, ws.xm, xm_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return bt_set_error(BT_ERR_HIP, "copy of the stage output");
    // (x3: not every operand-splitting kernel of the frontend raises the flag itself -- what they all do is turn an operand
    // beyond the fp16 range into inf / NaN, which the stage's exit looks for like the head does for the logits)
    if (x3) LAUNCH_CAT(CAT_HEAD, s, launch_finite_rows(ws.xm, (long)B * T, D, ws.status, s), "range check of the stage output");
    return BT_OK;
  }
  for (int l = 0; l < d.n_layers; ++l) {
    // (hl8 shadow for the next layer's QKV: only when that layer has the weights for it)
    const bool next_f8 = fast_x3 && ws.x3_gemm_fp8 >= 2 && l + 1 < d.n_layers && d.layers[l + 1].w_qkvg_f8 && d.layers[l].w_ff2_f8;
    int rc = fast_x3 ? run_layer_x3(pf, d.layers[l], d.rope, ws, B, T, d.ff_mult, s, xmb_f8, next_f8)
             : fast_layers ? run_layer_half(pf, d.layers[l], d.rope, ws, B, T, d.ff_mult, s)
                         : run_pair(pf, d.layers[l], d.rope, ws.xm, use_shadow ? ws.xmb : nullptr, ws, B, T, 1, 0, prec, s, nullptr,
                                    d.ff_mult, x3);
    if (rc) return rc;
    xmb_f8 = next_f8;
  }
  if (last == 1) {
    if (!d.norm_out_g) return bt_set_error(BT_ERR_ARG, "stage exit after transformer_blocks needs norm_out_g");
    LAUNCH_CAT(CAT_HEAD, s, launch_norm_out(ws.xm, d.norm_out_g, d_out, (long)B * T, D, s, x3 ? ws.status : nullptr), "final norm");
    return BT_OK;
  }
  HeadP hp;
  hp.x = ws.xm; hp.w = d.head_w; hp.b0 = d.head_b[0]; hp.b1 = d.head_b[1];
  hp.beat = d_beat; hp.downbeat = d_downbeat; hp.M = B * T; hp.D = D; hp.sum_head = d.sum_head; hp.prenorm = 0;
  hp.status = x3 ? ws.status : nullptr;
  LAUNCH_CAT(CAT_HEAD, s, launch_head(hp, s), "head");
  return BT_OK;
}

int bt_forward_unit(bt_engine* e, void* stream, int prec, int unit, int index, const float* d_in, float* d_out, int B, int T,
                    void* d_ws, size_t ws_bytes) {
  if (!e || !d_in || !d_out || !d_ws) return bt_set_error(BT_ERR_ARG, "null argument");
  const bt_model_desc& d = e->d;
  const int max_T = d.rope_len > 0 ? d.rope_len : 1536;
  if (B <= 0 || T <= 0 || T > max_T) return bt_set_error(BT_ERR_ARG, "need B >= 1 and 1 <= T <= bt_model_desc.rope_len");
  if (prec == BT_PREC_F32X3) prec = BT_PREC_F32;   // (sub-modules: the exact path)
  if (prec != BT_PREC_F32 && prec != BT_PREC_HALF) return bt_set_error(BT_ERR_ARG, "unknown precision");
  prof::State* pf = &e->prof;
  const int D = d.transformer_dim;
  Workspace ws = carve((char*)d_ws, B, T, D, d.ff_mult, prec);
  if (ws.total > ws_bytes) return bt_set_error(BT_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const bool block_unit = unit == BT_UNIT_PARTIAL || unit == BT_UNIT_CONV;
  const bool layer_unit = unit == BT_UNIT_ATTN || unit == BT_UNIT_FF;
  const bool leaf_unit = unit == BT_UNIT_FRONT_ATTN || unit == BT_UNIT_FRONT_FF;
  if (leaf_unit && (index < 0 || index > 5 || !d.partial_transformers)) return bt_set_error(BT_ERR_ARG, "partial-transformer leaf index out of range");
  if (block_unit && (index < 0 || index > 2)) return bt_set_error(BT_ERR_ARG, "frontend block index out of range");
  if (layer_unit && (index < 0 || index >= d.n_layers)) return bt_set_error(BT_ERR_ARG, "layer index out of range");
  auto copy_in = [&](size_t bytes) -> int {
    if (d_in != d_out && hipMemcpyAsync(d_out, d_in, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return bt_set_error(BT_ERR_HIP, "copy of the unit's input");
    return BT_OK;
  };
  switch (unit) {
    case BT_UNIT_STEM: {
      StemP sp;
      sp.spect = d_in; sp.x = d_out; sp.bn1_scale = d.bn1_scale; sp.bn1_shift = d.bn1_shift;
      sp.w = d.stem_w; sp.bias = d.stem_b; sp.B = B; sp.T = T;
      LAUNCH_CAT(CAT_STEM, s, launch_stem(sp, s), "stem");
      return BT_OK;
    }
    case BT_UNIT_PARTIAL: {
      if (!d.partial_transformers) return bt_set_error(BT_ERR_ARG, "this model has no partial transformers");
      const int F = 32 >> index;
      if (int rc = copy_in((size_t)B * T * 1024 * 4)) return rc;
      if (int rc = run_pair(pf, d.front[index][0], d.rope, d_out, nullptr, ws, B, T, F, 1, prec, s)) return rc;
      return run_pair(pf, d.front[index][1], d.rope, d_out, nullptr, ws, B, T, F, 2, prec, s);
    }
    case BT_UNIT_CONV: {
      const int C = 32 << index, F = 32 >> index;
      GemmP g;
      memset(&g, 0, sizeof g);
      g.A = d_in; g.W = d.conv_w[index][prec]; g.M = B * T * (F / 2); g.N = 2 * C; g.K = 6 * C;
      g.epi = GEMM_EPI_STORE; g.flags = GEMM_F_CONV | GEMM_F_A_F32 | GEMM_F_BIAS | GEMM_F_GELU | GEMM_F_OUT_F32;
      g.bias = d.conv_b[index]; g.out = d_out; g.ldo = 2 * C;
      g.conv_C2 = 2 * C; g.conv_T = T; g.conv_F = F / 2;
      LAUNCH_CAT(CAT_CONV, s, launch_gemm(g, prec, s), "frontend conv gemm");
      return BT_OK;
    }
    case BT_UNIT_LINEAR: {
      GemmP g;
      memset(&g, 0, sizeof g);
      g.A = d_in; g.lda = 1024; g.W = d.lin_w[prec]; g.M = B * T; g.N = D; g.K = 1024;
      g.epi = GEMM_EPI_STORE; g.flags = GEMM_F_A_F32 | GEMM_F_BIAS | GEMM_F_OUT_F32;
      g.bias = d.lin_b; g.out = d_out; g.ldo = D;
      LAUNCH_CAT(CAT_LINEAR, s, launch_gemm(g, prec, s), "frontend linear gemm");
      return BT_OK;
    }
    case BT_UNIT_ATTN:
    case BT_UNIT_FF: {
      if (int rc = copy_in((size_t)B * T * D * 4)) return rc;
      return run_pair(pf, d.layers[index], d.rope, d_out, nullptr, ws, B, T, 1, 0, prec, s, nullptr, d.ff_mult, false,
                      unit == BT_UNIT_ATTN ? 1 : 2);
    }
    case BT_UNIT_FRONT_ATTN:
    case BT_UNIT_FRONT_FF: {
      // a leaf of a partial transformer on its own: [sequences, tokens, C] rows like a main layer's (mode 0), generic kernels
      const bt_pair_weights& pw = d.front[index >> 1][index & 1];
      if (int rc = copy_in((size_t)B * T * pw.dim * 4)) return rc;
      return run_pair(pf, pw, d.rope, d_out, nullptr, ws, B, T, 1, 0, prec, s, nullptr, 4, false, unit == BT_UNIT_FRONT_ATTN ? 1 : 2);
    }
    case BT_UNIT_NORM:
      if (!d.norm_out_g) return bt_set_error(BT_ERR_ARG, "bt_model_desc.norm_out_g is not set");
      LAUNCH_CAT(CAT_HEAD, s, launch_norm_out(d_in, d.norm_out_g, d_out, (long)B * T, D, s), "final norm");
      return BT_OK;
    default:
      return bt_set_error(BT_ERR_ARG, "unknown unit");
  }
}

int bt_split_chunks(void* stream, const float* d_spect, int64_t n_frames, const int32_t* d_starts, int B, int T,
                    float* d_chunks) {
  if (!d_spect || !d_starts || !d_chunks || B <= 0 || T <= 0 || n_frames <= 0)
    return bt_set_error(BT_ERR_ARG, "bad argument to bt_split_chunks");
  LAUNCH(launch_split(d_spect, n_frames, d_starts, nullptr, B, T, d_chunks, (hipStream_t)stream), "split");
  return BT_OK;
}

int bt_split_chunks_batch(void* stream, const float* d_spect, const int32_t* d_chunk_table, int B, int T, float* d_chunks) {
  if (!d_spect || !d_chunk_table || !d_chunks || B <= 0 || T <= 0)
    return bt_set_error(BT_ERR_ARG, "bad argument to bt_split_chunks_batch");
  LAUNCH(launch_split(d_spect, 0, nullptr, d_chunk_table, B, T, d_chunks, (hipStream_t)stream), "split (batch)");
  return BT_OK;
}

int bt_aggregate(void* stream, const float* cb, const float* cd, const int32_t* d_starts, int B, int T, int border,
                 int64_t n_frames, float* d_beat, float* d_downbeat) {
  if (!cb || !cd || !d_starts || !d_beat || !d_downbeat || B <= 0 || T <= 0 || n_frames <= 0 || border < 0)
    return bt_set_error(BT_ERR_ARG, "bad argument to bt_aggregate");
  LAUNCH(launch_aggregate(cb, cd, d_starts, nullptr, nullptr, 1, B, T, border, n_frames, d_beat, d_downbeat,
                          (hipStream_t)stream), "aggregate");
  return BT_OK;
}

int bt_aggregate_batch(void* stream, const float* cb, const float* cd, const int32_t* d_chunk_table, const int32_t* d_pieces,
                       int n_pieces, int64_t max_frames, int T, int border, float* d_beat, float* d_downbeat) {
  if (!cb || !cd || !d_chunk_table || !d_pieces || !d_beat || !d_downbeat || n_pieces <= 0 || T <= 0 || max_frames <= 0 ||
      border < 0)
    return bt_set_error(BT_ERR_ARG, "bad argument to bt_aggregate_batch");
  LAUNCH(launch_aggregate(cb, cd, nullptr, d_chunk_table, d_pieces, n_pieces, 0, T, border, max_frames, d_beat, d_downbeat,
                          (hipStream_t)stream), "aggregate (batch)");
  return BT_OK;
}

static void fill_logmel(LogmelP& p, const bt_logmel_tables* t, float* d_spect) {
  p.window = t->window; p.twiddle = t->twiddle; p.mel_start = t->mel_start; p.mel_len = t->mel_len; p.mel_w = t->mel_w;
  p.spect = d_spect;
}

int bt_logmel(void* stream, const bt_logmel_tables* t, const float* d_audio, int64_t n_samples, float* d_spect) {
  if (!t || !d_audio || !d_spect) return bt_set_error(BT_ERR_ARG, "null argument");
  if (n_samples <= 512) return bt_set_error(BT_ERR_ARG, "signal too short: reflect padding needs more than 512 samples");
  LogmelP p;
  fill_logmel(p, t, d_spect);
  p.one = bt_span_t{d_audio, (long)n_samples, 0, (long)(1 + n_samples / 441)};
  p.tracks = nullptr; p.n_tracks = 1; p.max_frames = p.one.n_out;
  LAUNCH(launch_logmel(p, (hipStream_t)stream), "logmel");
  return BT_OK;
}

int bt_logmel_batch(void* stream, const bt_logmel_tables* t, const bt_span* d_tracks, int n_tracks, int64_t max_frames,
                    float* d_spect) {
  if (!t || !d_tracks || !d_spect || n_tracks <= 0 || max_frames <= 0)
    return bt_set_error(BT_ERR_ARG, "bad argument to bt_logmel_batch");
  static_assert(sizeof(bt_span) == sizeof(bt_span_t), "bt_span layout");
  LogmelP p;
  fill_logmel(p, t, d_spect);
  p.one = bt_span_t{nullptr, 0, 0, 0};
  p.tracks = reinterpret_cast<const bt_span_t*>(d_tracks); p.n_tracks = n_tracks; p.max_frames = (long)max_frames;
  LAUNCH(launch_logmel(p, (hipStream_t)stream), "logmel (batch)");
  return BT_OK;
}

int bt_resample(void* stream, const float* d_in, int64_t n_in, int up, int down, const float* d_filter, int half_len,
                float* d_out, int64_t n_out) {
  if (!d_in || !d_filter || !d_out || n_in <= 0 || n_out <= 0 || up <= 0 || down <= 0 || half_len < 0)
    return bt_set_error(BT_ERR_ARG, "bad argument to bt_resample");
  if (n_out > (n_in * up + down - 1) / down) return bt_set_error(BT_ERR_ARG, "n_out exceeds ceil(n_in * up / down)");
  const bt_span_t one{d_in, (long)n_in, 0, (long)n_out};
  LAUNCH(launch_resample(one, nullptr, 1, (long)n_out, up, down, d_filter, half_len, d_out, (hipStream_t)stream), "resample");
  return BT_OK;
}

int bt_resample_batch(void* stream, const bt_span* d_tracks, int n_tracks, int64_t max_n_out, int up, int down,
                      const float* d_filter, int half_len, float* d_out) {
  if (!d_tracks || !d_filter || !d_out || n_tracks <= 0 || max_n_out <= 0 || up <= 0 || down <= 0 || half_len < 0)
    return bt_set_error(BT_ERR_ARG, "bad argument to bt_resample_batch");
  const bt_span_t none{nullptr, 0, 0, 0};
  LAUNCH(launch_resample(none, reinterpret_cast<const bt_span_t*>(d_tracks), n_tracks, (long)max_n_out, up, down, d_filter,
                         half_len, d_out, (hipStream_t)stream), "resample (batch)");
  return BT_OK;
}

int bt_peaks(void* stream, const float* d_logits, int64_t n, int n_arrays, int32_t* d_idx, int32_t* d_count) {
  if (!d_logits || !d_idx || !d_count || n <= 0 || n_arrays <= 0) return bt_set_error(BT_ERR_ARG, "bad argument");
  LAUNCH(launch_peaks(d_logits, n, nullptr, n_arrays, d_idx, d_count, (hipStream_t)stream), "peaks");
  return BT_OK;
}

int bt_peaks_batch(void* stream, const float* d_logits, const int32_t* d_spans, int n_arrays, int32_t* d_idx,
                   int32_t* d_count) {
  if (!d_logits || !d_spans || !d_idx || !d_count || n_arrays <= 0)
    return bt_set_error(BT_ERR_ARG, "bad argument to bt_peaks_batch");
  LAUNCH(launch_peaks(d_logits, 0, d_spans, n_arrays, d_idx, d_count, (hipStream_t)stream), "peaks (batch)");
  return BT_OK;
}

int bt_peaks_host(const float* logits, int64_t n, int32_t* idx, int32_t* count) {
  // x == max_pool1d(x, 7, 1, 3) (implicit -inf padding) and x > 0 (postprocessor.py:93-99) for logits in HOST memory
  if (!logits || !idx || !count || n < 0) return bt_set_error(BT_ERR_ARG, "bad argument to bt_peaks_host");
  int32_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    const float v = logits[i];
    if (!(v > 0.0f)) continue;
    bool peak = true;
    for (int64_t j = std::max<int64_t>(0, i - 3); j <= std::min<int64_t>(n - 1, i + 3); ++j)
      if (logits[j] > v) { peak = false; break; }
    if (peak) idx[m++] = (int32_t)i;
  }
  *count = m;
  return BT_OK;
}

static int dedup_host(const int32_t* idx, int n, double* out, double width = 1.0) {
  // running-mean merge of frames not more than 1 apart (postprocessor.py:176-197)
  if (n <= 0) return 0;
  int m = 0;
  double mean = (double)idx[0];
  int count = 1;
  for (int i = 1; i < n; ++i) {
    double nxt = (double)idx[i];
    if (nxt - mean <= width) {
      ++count;
      mean += (nxt - mean) / count;
    } else {
      out[m++] = mean;
      mean = nxt;
      count = 1;
    }
  }
  out[m++] = mean;
  return m;
}

int bt_deduplicate_peaks_host(const int32_t* idx, int n, double width, double* out, int32_t* n_out) {
  if ((n > 0 && (!idx || !out)) || !n_out || n < 0) return bt_set_error(BT_ERR_ARG, "bad argument to bt_deduplicate_peaks_host");
  *n_out = dedup_host(idx, n, out, width);
  return BT_OK;
}

int bt_postprocess_host(const int32_t* beat_idx, int nb, const int32_t* down_idx, int nd, double fps, double* beats,
                        int32_t* n_beats, double* downbeats, int32_t* n_downbeats) {
  if ((nb > 0 && (!beat_idx || !beats)) || (nd > 0 && (!down_idx || !downbeats)) || !n_beats || !n_downbeats ||
      !(fps > 0))
    return bt_set_error(BT_ERR_ARG, "bad argument to bt_postprocess_host");
  int mb = dedup_host(beat_idx, nb, beats);
  for (int i = 0; i < mb; ++i) beats[i] = beats[i] / fps;
  int md = dedup_host(down_idx, nd, downbeats);
  for (int i = 0; i < md; ++i) downbeats[i] = downbeats[i] / fps;
  if (mb > 0) {
    // np.argmin(|beats - d|), first minimum wins (postprocessor.py:131-133).  beats ascend strictly (means of disjoint
    // runs), |b - d| is V-shaped over them and floating-point rounding is monotone, so the minimum sits at the first beat
    // >= d or at its predecessor -- the predecessor on an exact tie (= the first minimum).  O(log n) per downbeat instead of
    // the reference's O(n): 2000 x 2000 candidates per 5-minute track were 2.5 ms of host time with noisy logits.
    for (int i = 0; i < md; ++i) {
      const double d = downbeats[i];
      const int j = (int)(std::lower_bound(beats, beats + mb, d) - beats);
      int bi = j < mb ? j : mb - 1;
      if (j > 0 && (j >= mb || std::fabs(beats[j - 1] - d) <= std::fabs(beats[j] - d))) bi = j - 1;
      downbeats[i] = beats[bi];
    }
  }
  std::sort(downbeats, downbeats + md);  // np.unique
  md = (int)(std::unique(downbeats, downbeats + md) - downbeats);
  *n_beats = mb;
  *n_downbeats = md;
  return BT_OK;
}

// ---- Audio2Beats for one track in one call -------------------------------------------------------------------------------
int bt_audio2beats_plan(const bt_engine* e, int64_t n_in, int up, int down, int prec, bt_a2b_plan* plan) {
  if (!e || !plan || n_in <= 0 || up <= 0 || down <= 0) return bt_set_error(BT_ERR_ARG, "bad argument to bt_audio2beats_plan");
  if (prec != BT_PREC_F32 && prec != BT_PREC_HALF && prec != BT_PREC_F32X3) return bt_set_error(BT_ERR_ARG, "unknown precision");
  const int64_t n22 = up == down ? n_in : (n_in * up + down - 1) / down;
  if (n22 <= 512) return bt_set_error(BT_ERR_ARG, "signal too short: reflect padding needs more than 512 samples");
  const int64_t n = 1 + n22 / 441;
  const int chunk = 1500, border = 6, fresh = chunk - 2 * border;
  if (n + chunk >= 0x7fffffffL) return bt_set_error(BT_ERR_ARG, "track too long for 32-bit frame indices");
  memset(plan, 0, sizeof *plan);
  plan->n22 = n22; plan->n_frames = n;
  plan->T = n > fresh ? chunk : (int)n + 2 * border;
  plan->B = (int)((n + fresh - 1) / fresh);
  plan->result_words = 2 * n + 3;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  // what the forward's launches address comes FIRST, at offsets that depend on (B, T, precision) only: a captured forward is
  // then valid for every track of the same chunk count on the same allocation, whatever its exact length
  plan->forward_bytes = bt_workspace_bytes(e, plan->B, plan->T, prec);
  plan->off_forward = take(plan->forward_bytes);
  plan->off_chunks = take((size_t)plan->B * plan->T * 128 * 4);
  plan->off_chunk_logits = take((size_t)2 * plan->B * plan->T * 4);
  plan->off_wave22 = take(up == down ? 0 : (size_t)n22 * 4);
  plan->off_spect = take((size_t)n * 128 * 4);
  plan->off_logits = take((size_t)2 * n * 4);
  plan->off_result = take((size_t)plan->result_words * 4);
  plan->ws_bytes = off;
  return BT_OK;
}

int bt_audio2beats_enqueue(bt_engine* e, void* stream, int prec, const bt_logmel_tables* t, const float* d_audio, int64_t n_in,
                           int up, int down, const float* d_filter, int half_len, void* d_ws, size_t ws_bytes, int32_t* h_result,
                           int use_graph) {
  if (!e || !t || !d_audio || !d_ws || !h_result) return bt_set_error(BT_ERR_ARG, "null argument");
  if (up != down && (!d_filter || half_len < 0)) return bt_set_error(BT_ERR_ARG, "resampling needs the polyphase filter");
  bt_a2b_plan pl;
  if (int rc = bt_audio2beats_plan(e, n_in, up, down, prec, &pl)) return rc;
  if (pl.ws_bytes > ws_bytes) return bt_set_error(BT_ERR_WORKSPACE, "workspace too small (bt_audio2beats_plan)");
  std::lock_guard<std::mutex> lock(e->mu);   // (host-side enqueueing only: ~50 us per call)
  const int max_T = e->d.rope_len > 0 ? e->d.rope_len : 1536;
  if (pl.T > max_T) return bt_set_error(BT_ERR_ARG, "chunk longer than the rotary table");
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)d_ws;
  float* wave22 = (float*)(ws + pl.off_wave22);
  float* spect = (float*)(ws + pl.off_spect);
  float* chunks = (float*)(ws + pl.off_chunks);
  float* cb = (float*)(ws + pl.off_chunk_logits);
  float* cd = cb + (size_t)pl.B * pl.T;
  float* beat = (float*)(ws + pl.off_logits);
  float* downb = beat + pl.n_frames;
  int32_t* res = (int32_t*)(ws + pl.off_result);
  void* fws = ws + pl.off_forward;
  const int border = 6;
  const float* a22 = d_audio;
  if (up != down) {
    const bt_span_t one{d_audio, (long)n_in, 0, (long)pl.n22};
    LAUNCH(launch_resample(one, nullptr, 1, (long)pl.n22, up, down, d_filter, half_len, wave22, s), "resample");
    a22 = wave22;
  }
  LogmelP lp;
  fill_logmel(lp, t, spect);
  lp.one = bt_span_t{a22, (long)pl.n22, 0, (long)pl.n_frames};
  lp.tracks = nullptr; lp.n_tracks = 1; lp.max_frames = (long)pl.n_frames;
  LAUNCH(launch_logmel(lp, s), "logmel");
  LAUNCH(launch_split(spect, (long)pl.n_frames, nullptr, nullptr, pl.B, pl.T, chunks, s, border), "split");
  // ---- the forward: plain launches, or the replay of a graph captured here -------------------------------------------------
  const int warm_key = prec << 8 | pl.B;
  const bool graphable = use_graph && pl.T == 1500 && pl.B <= 16 && !e->prof.on &&
                         std::find(e->warm.begin(), e->warm.end(), warm_key) != e->warm.end();
  FwdGraph* g = nullptr;
  if (graphable) {
    for (auto& c : e->graphs)
      if (c.B == pl.B && c.T == pl.T && c.prec == prec && c.ws == d_ws) g = &c;
    if (!g) {
      // recorded on a private stream (thread-local capture mode: other threads' HIP calls are not affected); nothing runs
      hipGraph_t graph = nullptr;
      if (!e->cap_stream && hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking) != hipSuccess) e->cap_stream = nullptr;
      if (e->cap_stream && hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        const int rc = bt_forward_stages(e, e->cap_stream, prec, 0, 2, chunks, pl.B, pl.T, fws, pl.forward_bytes, nullptr, cb, cd);
        const hipError_t he = hipStreamEndCapture(e->cap_stream, &graph);
        hipGraphExec_t exec = nullptr;
        if (rc == BT_OK && he == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
          if (e->graphs.size() >= 8) {   // least recently used goes
            size_t old = 0;
            for (size_t i = 1; i < e->graphs.size(); ++i)
              if (e->graphs[i].stamp < e->graphs[old].stamp) old = i;
            (void)hipGraphExecDestroy(e->graphs[old].exec);
            e->graphs.erase(e->graphs.begin() + old);
          }
          e->graphs.push_back(FwdGraph{pl.B, pl.T, prec, d_ws, exec, 0});
          g = &e->graphs.back();
        }
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();   // (a failed capture must not poison the plain launches below)
      }
    }
  }
  if (g) {
    g->stamp = ++e->graph_clock;
    if (hipGraphLaunch(g->exec, s) != hipSuccess) return bt_set_error(BT_ERR_HIP, "replay of the captured forward");
  } else {
    if (int rc = bt_forward_stages(e, s, prec, 0, 2, chunks, pl.B, pl.T, fws, pl.forward_bytes, nullptr, cb, cd)) return rc;
    if (pl.T == 1500 && pl.B <= 16 && std::find(e->warm.begin(), e->warm.end(), warm_key) == e->warm.end()) e->warm.push_back(warm_key);
  }
  LAUNCH(launch_aggregate(cb, cd, nullptr, nullptr, nullptr, 1, pl.B, pl.T, border, (long)pl.n_frames, beat, downb, s), "aggregate");
  LAUNCH(launch_peaks(beat, (long)pl.n_frames, nullptr, 2, res, res + 2 * pl.n_frames, s), "peaks");
  // the range flag of the forward (first word of ITS workspace; zero for the other precisions) rides behind the counts
  if (prec == BT_PREC_F32X3) {
    if (hipMemcpyAsync(res + 2 * pl.n_frames + 2, fws, 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return bt_set_error(BT_ERR_HIP, "copy of the range flag");
  } else if (hipMemsetAsync(res + 2 * pl.n_frames + 2, 0, 4, s) != hipSuccess) {
    return bt_set_error(BT_ERR_HIP, "clearing the range flag");
  }
  if (hipMemcpyAsync(h_result, res, (size_t)pl.result_words * 4, hipMemcpyDeviceToHost, s) != hipSuccess)
    return bt_set_error(BT_ERR_HIP, "device-to-host copy of the result");
  return BT_OK;
}

void bt_profile_begin(bt_engine* e) {
  if (!e) return;
  for (auto& r : e->prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  e->prof.recs.clear();
  e->prof.on = true;
}

int bt_profile_end(bt_engine* e, double* ms_by_category, int32_t* launches_by_category, int n_categories) {
  if (!e) return bt_set_error(BT_ERR_ARG, "null engine");
  e->prof.on = false;
  if (!ms_by_category || !launches_by_category || n_categories < CAT_COUNT)
    return bt_set_error(BT_ERR_ARG, "need room for BT_PROFILE_CATEGORIES entries");
  for (int i = 0; i < n_categories; ++i) { ms_by_category[i] = 0.0; launches_by_category[i] = 0; }
  for (auto& r : e->prof.recs) {
    hipError_t e = hipEventSynchronize(r.b);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.a, r.b);
    if (e != hipSuccess) return bt_set_error(BT_ERR_HIP, hipGetErrorString(e));
    ms_by_category[r.cat] += ms;
    launches_by_category[r.cat] += 1;
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  e->prof.recs.clear();
  return BT_OK;
}

int bt_ff_fused(void* stream, int prec, const bt_pair_weights* w, float* d_x, int64_t M) {
  if (!w || !d_x || M <= 0 || w->dim > 128 || !w->w_ff_frag[prec]) return bt_set_error(BT_ERR_ARG, "bad argument to bt_ff_fused");
  FusedFFP ff;
  ff.x = d_x; ff.M = M; ff.C = w->dim; ff.wfrag = w->w_ff_frag[prec]; ff.b1 = w->b_ff1; ff.b2 = w->b_ff2;
  ff.xb = nullptr;
  LAUNCH(launch_ff_fused(ff, prec, (hipStream_t)stream), "fused feed-forward");
  return BT_OK;
}

int bt_gemm(void* stream, int prec, const bt_gemm_args* a) {
  if (!a) return bt_set_error(BT_ERR_ARG, "null argument");
  GemmP g;
  memset(&g, 0, sizeof g);
  g.A = a->A; g.lda = a->lda; g.W = a->W; g.M = a->M; g.N = a->N; g.K = a->K; g.epi = a->epi; g.flags = a->flags;
  g.bias = a->bias; g.out = a->out; g.ldo = a->ldo; g.x = a->x; g.ldx = a->ldx;
  g.conv_C2 = a->conv_C2; g.conv_T = a->conv_T; g.conv_F = a->conv_F;
  g.gates = a->gates; g.inner = a->inner; g.heads = a->heads; g.rope = a->rope;
  g.pdiv = a->pdiv; g.pmod = a->pmod; g.map_T = a->map_T; g.map_F = a->map_F;
  LAUNCH(launch_gemm(g, prec, (hipStream_t)stream), "gemm");
  return BT_OK;
}

int bt_attention(void* stream, int prec, const bt_attn_args* a) {
  if (!a) return bt_set_error(BT_ERR_ARG, "null argument");
  AttnP p;
  memset(&p, 0, sizeof p);
  p.qkv = a->qkv; p.ld = a->ld; p.gates = a->gates; p.out = a->out; p.n_seq = a->n_seq; p.L = a->L;
  p.heads = a->heads; p.inner = a->inner; p.o_div = a->o_div; p.o_outer = a->o_outer; p.o_inner = a->o_inner;
  p.o_tok = a->o_tok;
  LAUNCH(launch_attn_flash(p, prec, (hipStream_t)stream), "attention (flash)");
  return BT_OK;
}

int bt_outff_fused(void* stream, int prec, const bt_pair_weights* w, const void* d_ao, float* d_x, int64_t M, void* d_xb) {
  if (prec != BT_PREC_F32 && prec != BT_PREC_HALF && prec != BT_PREC_F32X3) return bt_set_error(BT_ERR_ARG, "unknown precision");
  const void* wf = !w ? nullptr : prec == BT_PREC_F32X3 ? w->w_outff_frag_x3 : w->w_outff_frag[prec];
  if (!w || !d_ao || !d_x || M <= 0 || w->dim > 128 || !wf) return bt_set_error(BT_ERR_ARG, "bad argument to bt_outff_fused");
  FusedOutFFP f;
  f.x = d_x; f.M = M; f.C = w->dim; f.ao = d_ao; f.wfrag = wf; f.b1 = w->b_ff1; f.b2 = w->b_ff2;
  f.xb = prec == BT_PREC_F32 ? nullptr : d_xb; f.abl = 0;
  LAUNCH(launch_outff_fused(f, prec, (hipStream_t)stream), "fused out-projection + feed-forward");
  return BT_OK;
}

int bt_attnff_fused(void* stream, int prec, const bt_pair_weights* w, const float* d_rope, float* d_x, int64_t M) {
  if (prec != BT_PREC_F32 && prec != BT_PREC_HALF && prec != BT_PREC_F32X3) return bt_set_error(BT_ERR_ARG, "unknown precision");
  const void* wf = !w ? nullptr : prec == BT_PREC_F32X3 ? w->w_attnff_frag_x3 : w->w_attnff_frag[prec];
  if (!w || !d_x || !d_rope || M <= 0 || w->dim > 128 || !wf) return bt_set_error(BT_ERR_ARG, "bad argument to bt_attnff_fused");
  FusedAttnFFP f;
  f.x = d_x; f.M = M; f.C = w->dim; f.b_gates = w->b_gates; f.rope = d_rope; f.wfrag = wf;
  f.b1 = w->b_ff1; f.b2 = w->b_ff2;
  LAUNCH(launch_attnff_fused(f, prec, (hipStream_t)stream), "fused frequency attention + feed-forward");
  return BT_OK;
}

int bt_layer_tail(void* stream, const bt_pair_weights* w, int hidden, const void* d_ao, float* d_x, int64_t M, void* d_xb,
                  float* d_ssq_out) {
  if (!w || !w->w_tail_frag || !d_ao || !d_x || M <= 0 || !layer_tail_supported(w->dim, hidden))
    return bt_set_error(BT_ERR_ARG, "bad argument to bt_layer_tail");
  LayerTailP t;
  t.x = d_x; t.M = M; t.C = w->dim; t.hidden = hidden; t.ao = d_ao; t.wfrag = w->w_tail_frag; t.b1 = w->b_ff1;
  t.b2 = w->b_ff2; t.xb = d_xb; t.ssq_out = d_ssq_out;
  LAUNCH(launch_layer_tail(t, (hipStream_t)stream), "layer tail");
  return BT_OK;
}

int bt_gemm3(void* stream, const bt_gemm3_args* a) {
  if (!a || !a->A || !a->W) return bt_set_error(BT_ERR_ARG, "null argument");
  Gemm3P g;
  memset(&g, 0, sizeof g);
  g.A = a->A; g.lda = a->lda; g.M = a->M; g.K = a->K; g.W = a->W; g.N = a->N; g.epi = a->epi; g.bias = a->bias;
  g.ssq_in = a->ssq_in; g.ssq_parts = a->ssq_parts; g.out = a->out; g.ldo = a->ldo; g.x = a->x; g.ldx = a->ldx;
  g.xb = a->xb; g.ssq_out = a->ssq_out; g.n_seq = a->n_seq; g.L = a->L; g.nblk = (a->L + 31) / 32; g.nbp = a->nbp;
  g.heads = a->heads; g.inner = a->heads * 32; g.rope = a->rope; g.qf = a->qf; g.kf = a->kf; g.vf = a->vf;
  g.gates = a->gates; g.b_gates = a->b_gates;
  g.x3 = a->x3; g.status = a->status;
  g.no_resid = a->no_resid; g.gelu = a->gelu; g.conv_C2 = a->conv_C2; g.conv_T = a->conv_T; g.conv_F = a->conv_F;
  if (!gemm3_supported(g)) return bt_set_error(BT_ERR_ARG, "shape not supported by bt_gemm3");
  LAUNCH(launch_gemm3(g, (hipStream_t)stream), "gemm3");
  return BT_OK;
}

int bt_gemm_mx8(void* stream, const void* d_A, const void* d_SA, const void* d_W, const void* d_SW, float* d_out, int M, int N, int K,
                int64_t ldo) {
  GemmMx8P g;
  g.A = d_A; g.SA = d_SA; g.W = d_W; g.SW = d_SW; g.out = d_out; g.ldo = (long)ldo; g.M = M; g.N = N; g.K = K;
  if (!d_A || !d_SA || !d_W || !d_SW || !d_out || !gemm_mx8_supported(g)) return bt_set_error(BT_ERR_ARG, "bad argument to bt_gemm_mx8");
  LAUNCH(launch_gemm_mx8(g, (hipStream_t)stream), "MX e4m3 gemm");
  return BT_OK;
}

int bt_attn_frag_blocks(int L) { return L > 0 ? attn_frag_blocks(L) : 0; }

int bt_attention_frag(void* stream, const bt_attn_frag_args* a) {
  if (!a || !a->q || !a->k || !a->v || !a->gates || !a->out) return bt_set_error(BT_ERR_ARG, "null argument");
  AttnFragP p;
  memset(&p, 0, sizeof p);
  p.q = a->q; p.k = a->k; p.v = a->v; p.gates = a->gates; p.out = a->out; p.n_seq = a->n_seq; p.L = a->L;
  p.heads = a->heads; p.inner = a->inner; p.nbp = a->nbp; p.o_div = a->o_div; p.o_outer = a->o_outer;
  p.o_inner = a->o_inner; p.o_tok = a->o_tok;
  p.x3 = a->x3; p.out_f32 = a->out_f32; p.status = a->status; p.fix_mask = a->scratch;
  if (p.x3 > 0 && !p.fix_mask) return bt_set_error(BT_ERR_ARG, "bt_attention_frag: x3 needs the scratch words (overflow map)");
  LAUNCH(launch_attn_frag(p, (hipStream_t)stream), "attention (fragment-major)");
  return BT_OK;
}

int bt_qkv_front(void* stream, int prec, const bt_pair_weights* w, const float* d_rope, const float* d_x, int B, int T, int F,
                 void* d_q, void* d_k, void* d_v, float* d_gates, int nbp) {
  if (prec != BT_PREC_HALF && prec != BT_PREC_F32X3) return bt_set_error(BT_ERR_ARG, "bt_qkv_front: BT_PREC_HALF or BT_PREC_F32X3");
  const bool x3 = prec == BT_PREC_F32X3;
  const void* wf = !w ? nullptr : x3 ? w->w_qkv_frag_x3 : w->w_qkv_frag;
  if (!w || !wf || !d_rope || !d_x || !d_q || !d_k || !d_v || !d_gates || w->dim > 128)
    return bt_set_error(BT_ERR_ARG, "bad argument to bt_qkv_front");
  QkvFrontP p;
  memset(&p, 0, sizeof p);
  p.x = d_x; p.B = B; p.T = T; p.F = F; p.C = w->dim; p.wfrag = wf; p.b_gates = w->b_gates; p.rope = d_rope;
  p.q = d_q; p.k = d_k; p.v = d_v; p.gates = d_gates; p.nbp = nbp; p.x3 = x3;
  LAUNCH(launch_qkv_front(p, (hipStream_t)stream), "frontend qkv projection");
  return BT_OK;
}

}  // extern "C"
