// Host-side launch plan of the v3 generator forward (reference
// models/networks_iid_hlgvit_crs_gd4_cfs_v3.py:392-1020): a fixed sequence of kernel launches over a
// caller-provided workspace.  No allocation, no synchronisation: the whole forward can be captured
// into a hipGraph by the caller.
//
// Workspace layout = one NHWC buffer per top-level stage output (so parity tests can read every
// stage of SURVEY Appendix D after a forward) + token scratch shared by all 24 transformer blocks.
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/cfen_hip.h"
#include "cfen_common.hpp"
#include "cfen_conv.hpp"
#include "cfen_internal.hpp"
#include "cfen_mlp.hpp"
#include "cfen_lvit.hpp"

int& cfen_tune_skip_classes() {
  static int v = 0;
  return v;
}
int& cfen_tune_gvit_dummy_wgs() { static int v = 0; return v; }
int& cfen_tune_gvit_dummy_us() { static int v = 100; return v; }
int& cfen_tune_gvit_dummy_stream() { static int v = 0; return v; }
int& cfen_tune_skip_from() { static int v = -1; return v; }   // what-if probe: the launches number skip_from .. skip_to of a forward are not launched (outputs invalid)
int& cfen_tune_skip_to() { static int v = -1; return v; }
int& cfen_tune_extra_launches() { static int v = 0; return v; }   // what-if probe: that many one-workgroup 1 us launches in front of every ViT block (what a launch costs a chain)
int& cfen_tune_gvit_dummy_levels() { static int v = 0; return v; }   // 1: the what-if probe replaces only the blocks that run as a launch per GEMM (levels 2-3)
int& cfen_tune_gvit_chain() { static int v = 1; return v; }   // only nets built with fragment-stream GViT weights (cfg.reserved bit 2) can use it
int& cfen_tune_gvit_stream() { static int v = 2; return v; }   // 0 never, 1 in the serial launch plan only, 2 (default, round 5) on every plan: which KERNELS produce the
                                                                // outputs no longer depends on the lane plan or on profiling (two-lane, serial and profiled forwards are bitwise equal;
                                                                // the stream kernels cost one forward at a time 2.80 -> 2.85 ms and gain 2.27 -> 2.24 with several in flight)
                                                                // (one forward at a time on the two-lane plan it is SLOWER, 2.85 against 2.80 ms: 277 us of latency against 134)
int& cfen_tune_tail_fused() { static int v = 2; return v; }   // 0 three launches, 1 ConvTranspose + 3x3 fused (round 4), 2 (default, round 5) the whole tail in one launch (k_tail.hip)
int& cfen_tune_up_fused() { static int v = 0; return v; }
int& cfen_tune_keep_stages() { static int v = 0; return v; }
int& cfen_tune_resblock_fused() { static int v = 0; return v; }   // 0 (default): MEASURED with three forwards in flight 2.44 against 2.48 ms -- the fused kernel (5-wave workgroups, 58 KB of LDS, 154 registers) is 13 us shorter alone and costs more CU-time beside other forwards
int& cfen_tune_head5() { static int v = 1; return v; }
int& cfen_tune_ln_fold() {
  static int v = 1;
  return v;
}
int& cfen_tune_fused_front_max_dim() {
  static int v = 192;
  return v;
}
int& cfen_tune_embed_gather() {
  static int v = 1;
  return v;
}
int& cfen_tune_lvit_window() {
  static int v = 1;
  return v;
}
int& cfen_tune_fold_in_gemm() {   // 1 (default): mlp_head.3's GEMM stores straight into the NHWC map (fold + Join2x2), no unpatchify launch
  static int v = 1;
  return v;
}
int& cfen_tune_attn_head_major() {
  static int v = 1;
  return v;
}
int& cfen_tune_head_fused() {   // 0 (default): three k_conv_tile launches.  MEASURED (MI355X, batch 8): k_head_fused moves 100 MB instead of 400 MB and is
  static int v = 0;             // SLOWER, 131 us against 108: with 3 input channels padded to 8 and 5 taps to 8 its MFMA work is 5x the algorithmic
  return v;                     // flops (22 % MFMA-busy, profiles/r03_*), and the unfused kernels already run at the HBM rate of their own maps
}
int& cfen_tune_stream_front() {   // k_front3 for the D = 384 LViT blocks: 0 never, 1 grouped decoder launches, 2 (default, round 4) always
  static int v = 2;             // (with three forwards in flight the single encoder instance on the stream kernels is 0.03 ms better: one whole-CU launch of 64 workgroups instead of 8 GEMM launches)
  return v;
}
int& cfen_tune_stream_mlp192() {   // LViT level 2 (D = 192) proj + MLP block: 1 (default, round 5) on k_mlp3 (fragment-stream weights, three-slot ring; mlp3.tm192 = 22: TWO 78 KB
  static int v = 1;                // workgroups a CU at 256 registers -- k_mlp2's occupancy without its two-stage ring, whose chunk period is one LDS-DMA issue -> landed
  return v;                        // latency: encoder 79.7 -> 56.4 us, grouped decoder launch 168.4 -> 128.2 us (961 TF), 2.148 -> 2.121 ms per step with four forwards in
}                                  // flight, profiles/r05_ab_stream_mlp192_two_wgs_per_cu.txt); 0: k_mlp2.  (k_mlp3<12, 3> on one 150 KB workgroup a CU, rounds 3-4: equal to k_mlp2.)
int& cfen_tune_stream_mlp() {   // k_mlp3 (k_stream.hip) for the D = 384 blocks: 0 never, 1 launches of >= 128 workgroups (at 512 x 512 the grouped
  static int v = 2;             // decoder launch; a single instance has 64 workgroups of 128 tokens: a quarter of the chip), 2 (default, round 4) always
  return v;
}

namespace {

struct Param {
  const void* ptr = nullptr;
  size_t need = 0;
};
struct Buf {
  size_t off = 0;
  int C = 0, cs = 0, H = 0, W = 0;
};
struct Vit {
  std::string name;
  bool global;
  int level, C, p, S, D, heads, hidden;
  // v5 LViT (networks_iid_hlgvit_crs_gd4_cfs_v5.py:1086-1106): the block runs on a quarter of the level's channels (6 / 12 / 24), between a 1x1
  // conv_shrink and conv_extend.  Those maps live at channel stride 8 / 16 / 24, so C, D count the zero slots of the padded token row;
  // Dn = real embedding dim (LayerNorm statistics), dh = head dim padded to 8 (real 6; the packer zero-fills and folds the softmax scale
  // into W_q), Da = heads * dh = width of q, k, v and of the attention output.  Everywhere else Dn = Da = D, dh = D / heads.
  int Dn, Da, dh, Cmap;
  bool shrink;
  int mapH;   // edge of the map the tokens tile (pooled edge for GViT)
  int ws;     // window edge on that map
  bool fused_mlp;   // LN2+FFN+mlp_head+fold run as one k_mlp launch
  bool fused_front; // gather+embedding+LN1+qkv run as one k_embed_qkv launch
  bool ln_fold1, ln_fold2;   // LN1 / LN2 ride on the qkv / ffn1 GEMM (k_gemm_dma row statistics + folded weights), no LayerNorm launch
  bool fused_window;// the whole block runs as one k_lvit_window launch (one workgroup per window)
  bool gstream;     // GViT with embedding dim 384: the block runs on the LViT-3 stream kernels (k_front3 / k_mlp3; "<name>.embed.ws" ... ".head.ws")
  bool chain;       // GViT: the GEMMs run as two persistent chains (k_gvit.hip) on fragment-stream weights ("<name>.embed.wf" ... ".head2.wf")
  bool stream_mlp;  // out_proj + LN2 + FFN + mlp_head + fold can run as one k_mlp3 launch on fragment-stream weights ("<name>.proj.ws" / ".ffn.ws" / ".head.ws")
};
struct ConvLayer {
  int kind, k, stride, pad, reflect, nsrc, Cin, Cin_real, Cout, Cout_pad, Kpad, nphase, ntaps, out_edge;
  bool tile;   // LDS-tiled kernel, weights "<layer>.wr" in the rows layout
  bool tz;     // Toeplitz 7x7 kernel (tails), weights "<layer>.wz"
};

inline int cs_of(int C) { return cfen_round_up(C, 8); }

}  // namespace

struct cfen_net {
  cfen_net_config cfg;
  int esz, KC;
  std::map<std::string, Param> params;
  std::map<std::string, Buf> bufs;
  std::map<std::string, ConvLayer> convs;
  std::vector<Vit> vits;
  size_t ws_bytes = 0;
  // Token scratch, one set per concurrently running transformer block (LViT / GViT of branch A / B)
  struct Scratch { size_t x0, x1, yn, qkv, att, hid, small, splitk, sync; };   // sync (GViT sets): grid-barrier words of the persistent chains (one per launch of a forward) + error word   // splitk: arrival counters + partial slabs of the split-K GEMMs (GViT sets)
  static constexpr size_t SPLITK_BYTES = 16u << 20;
  Scratch scr_set[6];
  size_t o_stats_set[3] = {0, 0, 0};
  // ActNorm2d layers whose parameters are still uninitialised (models/actnorm.py:25-37): the next EAGER forward runs the layer
  // raw (conv + bias), takes batch statistics, writes the folded epilogue table in place and the raw (weight, bias) pair to an_out
  // win > 0: the layer sits inside a v5 LViT module, which the reference calls once per window -- its first call (the one that
  // initialises) sees the top-left win x win window of every image only (v5:403-440)
  struct AnPending { const float* ones; const float* conv_bias; float* an_out; int win; };
  std::map<std::string, AnPending> an_pending;
  bool an_raw_pass = false;
  int output_f16 = 0;              // 1: xr / xs / xd are fp16 NCHW (round 6: what the sharded run gathers), written by the fused tail launch (k_tail.hip) only
  int output_u8 = 0;               // 1: xr / xs / xd are uint8 HWC (B,H,W,3) = util.tensor2im of the fp32 results, written by the tails' 7x7 launch (k_conv7_tz)
  int input_u8 = 0;                // 1: x is uint8 HWC (B,H,W,3), normalised to [-1,1] by the first launch (data/base_dataset.py:44-46)
  bool cfs = false;                // sibling generators networks_iid_hlgvit_crs_gd4_cfs.py / ..._crs_gd4.py (cfg.reserved bits 8..15 == 1 / 2): the three
                                   // levels run at the image's own resolution -- no ds_conv_e01 / us_conv_d01*, n_feats channels in head and tails
  bool crs = false;                // ..._crs_gd4.py (variant 2): D's skip fuse is a 1x1 conv over (D, R, S) upsampled maps instead of CFSM2G (crs:854,889)
  bool v5 = false;                 // ..._cfs_v5.py (variant 3): v3 with every LViT block between conv_shrink / conv_extend
  bool gvit_stream = false;        // cfg.reserved bit 2: GViT weights are also held as fragment streams (packing.pack_stream_tiles) -> persistent chains
  int gv_launch = 0;               // persistent-chain launches enqueued so far in this forward (each takes its own barrier word)
  static constexpr int GV_SYNC_WORDS = 1024, GV_ERR_WORD = 512;
  std::set<std::string> gvit_low;  // GViT outputs the last forward left at low resolution (x4 bilinear inside the fuse conv, ConvDesc::up4): cfen_net_stage refuses them
  bool gv_skip_up = false;         // run_vit_g (GViT): leave the block's result in the low-resolution scratch map, no k_upsample4
  bool tail_whole = false;         // ... and the tails' 3x3 outputs too (k_tail_fused)
  bool stages_on_chip = false;     // the last forward kept the us_conv_d01* maps in LDS (k_up_conv3_fused without "net.keep_stages"): cfen_net_stage refuses them
  bool head5 = false;              // head.0.0 can run on k_head5 (reads the network input itself)
  bool wtile = false;              // cfg.reserved bit 1: GViT weights are packed tile-major (CfenGemmPtrs::wtile, packing.pack_wtile)
  size_t wbytes(const Vit& v, int N, int K) const { return (size_t)(v.global && wtile ? cfen_round_up(N, 96) : N) * K * esz; }
  int full = 0;                    // image edge
  int launch_idx = 0;              // launches of this forward so far ("net.skip_from" / "net.skip_to")
  int blk_kind = 0;                // 0 CNN, 1 GViT block, 2 LViT block (selects bits 8.. / 16.. of "net.skip_classes")
  unsigned char* base = nullptr;   // workspace of the current / last forward
  hipStream_t stream = nullptr;    // stream of the lane being enqueued
  // fork/join over internal side streams (captured into the caller's hipGraph like any other work):
  //   GViT runs beside LViT of the same level, the S decoder beside the R decoder.
  bool parallel = true;
  // every fork uses a stream that has not been forked before in this forward: re-forking a stream that
  // already joined makes hipStreamEndCapture (ROCm 7.2) recurse without bound
  static constexpr int NSIDE = 16;
  hipStream_t side[NSIDE] = {};
  int side_next = 0;
  hipStream_t fresh_side() { return side[side_next++ % NSIDE]; }
  std::vector<hipEvent_t> evs;
  size_t ev_next = 0;
  std::vector<hipGraphExec_t> execs;   // instantiated launch plans (cfen_net_graph_capture)
  int order(hipStream_t before, hipStream_t after) {   // work enqueued on `after` from now on waits for `before`'s work so far
    if (before == after) return CFEN_OK;
    if (CfenGraphRecorder* rec = cfen_recorder()) {     // building a graph: lanes are dependency lists, not streams
      std::vector<hipGraphNode_t>& a = rec->tail[after];
      for (hipGraphNode_t n : rec->tail[before])
        if (std::find(a.begin(), a.end(), n) == a.end()) a.push_back(n);
      return CFEN_OK;
    }
    if (ev_next == evs.size()) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { cfen_set_error("net: cannot create event"); return CFEN_ERR_HIP; }
      evs.push_back(e);
    }
    hipEvent_t e = evs[ev_next++];
    if (hipEventRecord(e, before) != hipSuccess || hipStreamWaitEvent(after, e, 0) != hipSuccess) {
      cfen_set_error("net: stream fork/join failed");
      return CFEN_ERR_HIP;
    }
    return CFEN_OK;
  }
  ~cfen_net() {
    for (hipGraphExec_t g : execs) if (g) (void)hipGraphExecDestroy(g);
    for (hipEvent_t e : evs) (void)hipEventDestroy(e);
    for (hipStream_t q : side) if (q) (void)hipStreamDestroy(q);
  }
  // optional per-launch timing (cfen_net_profile): one event pair per launch, tagged with a class
  bool profiling = false;
  struct Rec { int cls; double flops; hipEvent_t a, b; std::string label; double ms; std::string kernel; double bytes; };
  double prof_bytes = 0.0;               // algorithmic bytes of the launch about to be enqueued (weight-streaming GEMMs set it; consumed by prof_begin)
  std::vector<Rec> recs, last_profile;   // last_profile: per-launch detail of the latest cfen_net_profile
  std::string label;                     // what the launches being enqueued belong to (layer / block step)
  int prof_begin(int cls, double flops) {
    if (!profiling) return -1;
    Rec r; r.cls = cls; r.flops = flops; r.label = label; r.ms = 0; r.bytes = prof_bytes;
    prof_bytes = 0.0;
    cfen_kernel_log().clear();
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
    (void)hipEventRecord(r.a, stream);
    recs.push_back(r);
    return (int)recs.size() - 1;
  }
  void prof_end(int id) {
    if (id < 0) return;
    (void)hipEventRecord(recs[id].b, stream);
    recs[id].kernel = cfen_kernel_log();
  }

  size_t alloc(size_t bytes) {
    size_t off = ws_bytes;
    ws_bytes += (bytes + 255) / 256 * 256;
    return off;
  }
  void add_map(const std::string& n, int C, int edge) {
    Buf b;
    b.C = C; b.cs = cs_of(C); b.H = edge; b.W = edge;
    b.off = alloc((size_t)cfg.batch * edge * edge * b.cs * esz);
    bufs[n] = b;
  }
  void need(const std::string& n, size_t bytes) { params[n].need = bytes; }
  void add_conv(const std::string& n, int kind, int k, int stride, int pad, int reflect, int nsrc, int Cin_real, int Cout, int out_edge) {
    ConvLayer c;
    const int Cin = cs_of(Cin_real);
    c.kind = kind; c.k = k; c.stride = stride; c.pad = pad; c.reflect = reflect; c.nsrc = nsrc; c.Cin = Cin; c.Cout = Cout;
    c.Cin_real = Cin_real; c.out_edge = out_edge;
    c.Cout_pad = cfen_round_up(Cout, 16);
    c.nphase = kind == 1 ? 4 : 1;
    c.ntaps = kind == 1 ? 4 : k * k * nsrc;
    c.Kpad = cfen_round_up(c.ntaps * Cin, KC);
    if (kind == 1) {
      c.tile = cfen_convT_tile_supported(cfg.dtype, Cin, c.Cout_pad, out_edge / 2, out_edge / 2);
      if (c.tile) c.Kpad = cfen_convT_tile_kpad(cfg.dtype, Cin);
    } else {
      c.tile = cfen_conv_tile_supported(cfg.dtype, kind, k, stride, pad, nsrc, Cin, c.Cout_pad, out_edge, out_edge);
      if (c.tile) c.Kpad = cfen_conv_tile_kpad(cfg.dtype, k, Cin);
    }
    c.tz = n.size() > 6 && n.compare(n.size() - 6, 6, ".conv7") == 0 &&
           cfen_conv7_tz_supported(cfg.dtype, k, stride, pad, nsrc, Cin, Cout, 1, out_edge, out_edge);   // the tails write fp32 NCHW
    if (c.tz) { c.tile = false; c.Kpad = cfen_conv7_tz_kpad(); }
    convs[n] = c;
    need(n + (c.tz ? ".wz" : c.tile ? ".wr" : ".w"), (size_t)c.nphase * c.Cout_pad * c.Kpad * esz);
    need(n + ".scale", (size_t)c.Cout_pad * 4);
    need(n + ".shift", (size_t)c.Cout_pad * 4);
  }
  void* at(size_t off) const { return base + off; }
  void* map_ptr(const std::string& n) const { return base + bufs.at(n).off; }
  const void* P(const std::string& n) const { return params.at(n).ptr; }
  const float* Pf(const std::string& n) const { return (const float*)params.at(n).ptr; }

  int build();
  // one convolution of up to CFEN_MAX_GROUPS same-shaped layers (the R / S / D copies of a decoder layer) as ONE launch
  struct ConvCall {
    std::string layer, in0, in1, res0, res1, out; float* nchw_out = nullptr; std::string in2 = "";   // "" = absent
    const void* up_low = nullptr; int up_h = 0, up_w = 0, up_cs = 0;   // in1 is read as x4 of this low-resolution map (ConvDesc::up4)
  };
  int run_conv_g(int ng, const ConvCall* c, int act);
  int run_conv(const std::string& layer, const std::string& in0, const char* in1, const char* res0, const char* res1, int act,
               const std::string& out, float* nchw_out) {
    ConvCall c{layer, in0, in1 ? in1 : "", res0 ? res0 : "", res1 ? res1 : "", out, nchw_out};
    return run_conv_g(1, &c, act);
  }
  struct VitCall { const Vit* v; std::string in, out; };
  // first use of a workspace: its synchronisation words (split-K arrival counters, barrier and error words of the chains) start at zero.  Ordered on
  // the CALLER'S stream (a kernel launch on it) for an eager forward -- torch's lane streams are non-blocking, a memset on the legacy null stream is
  // not ordered against them -- and synchronously for a graph capture (a one-time set-up call without a stream of its own).  Keyed by
  // (address, size): the caller must not write into the workspace between forwards (include/cfen_hip.h), and hands a NEW net a workspace of its own.
  const void* primed_base = nullptr;
  size_t primed_bytes = 0;
  int prime_workspace(bool capturing) {
    if (base == primed_base && primed_bytes == ws_bytes) return CFEN_OK;
    for (int k = 1; k < 6; k += 2) {
      if (capturing) {
        if (hipMemset(at(scr_set[k].splitk), 0, CFEN_SPLITK_COUNTERS * sizeof(unsigned)) != hipSuccess ||
            hipMemset(at(scr_set[k].sync), 0, GV_SYNC_WORDS * sizeof(unsigned)) != hipSuccess) {
          cfen_set_error("net: cannot zero the synchronisation words of the workspace");
          return CFEN_ERR_HIP;
        }
      } else {
        if (int rc = cfen_zero_words_impl(at(scr_set[k].splitk), CFEN_SPLITK_COUNTERS, stream)) return rc;
        if (int rc = cfen_zero_words_impl(at(scr_set[k].sync), GV_SYNC_WORDS, stream)) return rc;
      }
    }
    if (capturing && hipDeviceSynchronize() != hipSuccess) {
      cfen_set_error("net: cannot zero the synchronisation words of the workspace (synchronize)");
      return CFEN_ERR_HIP;
    }
    primed_base = base;
    primed_bytes = ws_bytes;
    return CFEN_OK;
  }
  int run_vit_g(int ng, const VitCall* c, int scr0);   // group member g uses scratch set scr0 + 2 g
  int run_level_g(int ng, const char* tags, int l, const std::string* in, const char* extra_res, const std::string* out, hipStream_t sm,
                  hipStream_t sg);
  int run_decoder(float* const* outs, hipStream_t sm, hipStream_t sg);
  int forward(const float* x, float* xr, float* xs, float* xd);
};

#define TRY(expr)             \
  do {                        \
    int rc__ = (expr);        \
    if (rc__) return rc__;    \
  } while (0)

// launch `expr`, attributing its time to kernel class `cls` when profiling
#define TRYP(cls, flops, expr)            \
  do {                                    \
    if (cfen_tune_skip_classes() & ((1 << (cls)) | (1 << ((cls) + 8 * blk_kind)))) break; /* what-if timing: outputs are garbage */ \
    { const int li__ = launch_idx++; if (li__ >= cfen_tune_skip_from() && li__ <= cfen_tune_skip_to()) break; } \
    int id__ = prof_begin(cls, flops);    \
    int rc__ = (expr);                    \
    prof_end(id__);                       \
    if (rc__) return rc__;                \
  } while (0)

enum { K_GEMM = 0, K_ATTN = 1, K_LNORM = 2, K_TOKEN = 3, K_CONV = 4, K_NORM = 5, K_MLP = 6, K_NCLASS = 7 };

int cfen_net::build() {
  const int nf = cfg.n_feats, N = cfg.load_size, B = cfg.batch;
  esz = cfg.dtype == CFEN_F16 ? 2 : 4;
  KC = cfg.dtype == CFEN_F16 ? 32 : 16;
  CFEN_CHECK_ARG(cfg.dtype == CFEN_F16 || cfg.dtype == CFEN_F32, "net: unknown dtype %d", cfg.dtype);
  CFEN_CHECK_ARG(B > 0 && nf > 0 && nf % 8 == 0, "net: batch must be > 0 and n_feats a multiple of 8");
  CFEN_CHECK_ARG(N == 8 * cfg.patch_size, "net: loadSize (%d) must equal 8*patch_size (%d) (reference crop nesting, v3:403-529)", N,
                 cfg.patch_size);
  CFEN_CHECK_ARG(cfg.patch_size % 2 == 0 && N % 64 == 0, "net: loadSize must be a multiple of 64 (16-pixel tiles at level 3)");
  CFEN_CHECK_ARG(cfg.num_heads > 0 && (nf * 4) % cfg.num_heads == 0, "net: embedding dim not divisible by heads");
  CFEN_CHECK_ARG(4 * nf <= 128, "net: n_feats > 32 unsupported");
  CFEN_CHECK_ARG(cfg.hidden_dim_ratio > 0, "net: hidden_dim_ratio must be positive");
  const int variant = (cfg.reserved >> 8) & 0xff;
  CFEN_CHECK_ARG(variant >= 0 && variant <= 3, "net: unknown generator variant %d (0 = ..._cfs_v3, 1 = ..._cfs, 2 = ..._crs_gd4, 3 = ..._cfs_v5)", variant);
  cfs = variant == 1 || variant == 2;
  crs = variant == 2;
  v5 = variant == 3;
  wtile = (cfg.reserved & 2) != 0;
  gvit_stream = (cfg.reserved & 4) != 0 && cfg.dtype == CFEN_F16;
  CFEN_CHECK_ARG(!v5 || nf % cfg.num_heads == 0, "net (v5): shrunk embedding dim (n_feats) not divisible by heads");
  full = cfs ? N : 2 * N;

  // ---- transformer instances (reference v3:136-246) ----
  static const char* br = "rsd";
  auto lv = [&](const std::string& name, int l) {
    Vit v;
    v.name = name; v.global = false; v.level = l; v.C = nf << (l - 1); v.p = 2;
    v.ws = cfg.patch_size; v.mapH = N >> (l - 1);
    v.S = (v.ws / 2) * (v.ws / 2); v.D = v.C * 4; v.heads = cfg.num_heads << (l - 1); v.hidden = v.D * cfg.hidden_dim_ratio;
    v.Cmap = v.C; v.Dn = v.Da = v.D; v.dh = v.D / v.heads; v.shrink = false;
    if (v5) {
      const int Cq = v.C / 4;                      // v5:1095-1109
      v.shrink = true;
      v.C = cs_of(Cq); v.D = v.C * 4; v.Dn = Cq * 4; v.hidden = v.Dn * cfg.hidden_dim_ratio;
      v.dh = cfen_round_up(v.Dn / v.heads, 8); v.Da = v.heads * v.dh;
    }
    vits.push_back(v);
  };
  auto gv = [&](const std::string& name, int l) {
    Vit v;
    v.name = name; v.global = true; v.level = l; v.C = nf << (l - 1); v.p = 4;
    v.mapH = (N >> (l - 1)) / 4; v.ws = v.mapH;
    v.S = (v.mapH / 4) * (v.mapH / 4); v.D = v.C * 16; v.heads = cfg.num_heads << (l - 1);
    v.hidden = v.D * cfg.hidden_dim_ratio;
    v.Cmap = v.C; v.Dn = v.Da = v.D; v.dh = v.D / v.heads; v.shrink = false;
    if (name == "globalvit_encoder_02") v.hidden = v.C * 4 * cfg.hidden_dim_ratio;   // v3:200 quirk (patch_dim, not patch_dim*2)
    vits.push_back(v);
  };
  for (int l = 1; l <= 3; ++l) lv("localvit_encoder_0" + std::to_string(l), l);
  for (int b = 0; b < 3; ++b)
    for (int l = 3; l >= 1; --l) lv("localvit_decoder_0" + std::to_string(l) + br[b], l);
  for (int l = 1; l <= 3; ++l) gv("globalvit_encoder_0" + std::to_string(l), l);
  for (int b = 0; b < 3; ++b)
    for (int l = 3; l >= 1; --l) gv("globalvit_decoder_0" + std::to_string(l) + br[b], l);

  size_t max_md_l = 0, max_mh_l = 0, max_md_g = 0, max_mh_g = 0, max_small = 0;
  for (Vit& v : vits) {
    v.fused_mlp = !v.global && !v.shrink && cfen_mlp_supported(v.D, v.hidden, cfg.dtype);
    v.fused_front = !v.global && !v.shrink && cfen_embed_qkv_supported(v.D);
    v.fused_window = !v.global && v.fused_mlp && v.fused_front && cfen_lvit_window_supported(cfg.dtype, v.D, v.heads, v.S, v.hidden);
    v.stream_mlp = !v.global && !v.shrink && (v.D == 384 ? !v.fused_mlp : v.D == 192) && v.hidden <= 4 * v.D && cfen_mlp3_supported(cfg.dtype, v.D, v.hidden);
    v.gstream = v.global && !v.shrink && cfg.dtype == CFEN_F16 && v.D == 384 && v.hidden <= 4 * v.D && cfen_mlp3_supported(cfg.dtype, v.D, v.hidden) &&
                v.C % 8 == 0;
    v.chain = v.global && gvit_stream && !v.shrink && v.D % 128 == 0 && v.hidden % 128 == 0;
    v.ln_fold1 = !v.fused_front && v.Dn == v.D && (v.D * esz) % 128 == 0;
    v.ln_fold2 = !v.fused_mlp && v.Dn == v.D && (v.D * esz) % 128 == 0;
    CFEN_CHECK_ARG(v.Dn % v.heads == 0, "net: %s embedding dim %d not divisible by %d heads", v.name.c_str(), v.Dn, v.heads);
    CFEN_CHECK_ARG(v.mapH % v.ws == 0 && v.ws % v.p == 0 && v.S >= 1, "net: %s does not tile its map", v.name.c_str());
    const size_t ntok = (size_t)B * (v.mapH / v.ws) * (v.mapH / v.ws) * v.S;
    (v.global ? max_md_g : max_md_l) = std::max(v.global ? max_md_g : max_md_l, ntok * std::max(v.D, v.Da));
    (v.global ? max_mh_g : max_mh_l) = std::max(v.global ? max_mh_g : max_mh_l, ntok * v.hidden);
    if (v.global) max_small = std::max(max_small, (size_t)B * v.mapH * v.mapH * v.C);
    const std::string& n = v.name;
    need(n + ".embed.w", wbytes(v, v.D, v.D)); need(n + ".embed.b", (size_t)v.D * 4);
    if (v.fused_front) { need(n + ".embed.wk", (size_t)v.D * v.D * esz); need(n + ".qkv.wk", (size_t)3 * v.D * v.D * esz); }
    if (v.fused_window) {
      // the window kernel's fragment stream: embedding D x D, K / V 2D x D, per head 32 x D + D x 32, two MLP pairs of 2 x hidden x D
      need(n + ".lw.ws", ((size_t)3 * v.D * v.D + (size_t)v.heads * 64 * v.D + (size_t)4 * v.hidden * v.D) * esz);
    }
    need(n + ".pos", (size_t)v.S * v.D * esz);
    need(n + ".ln1.g", (size_t)v.D * 4); need(n + ".ln1.b", (size_t)v.D * 4);
    need(n + ".qkv.w", wbytes(v, 3 * v.Da, v.D));
    need(n + ".proj.w", wbytes(v, v.D, v.Da));
    if (v.ln_fold1) { need(n + ".qkv.wl", wbytes(v, 3 * v.Da, v.D)); need(n + ".qkv.s", (size_t)3 * v.Da * 4); need(n + ".qkv.bl", (size_t)3 * v.Da * 4); }
    if (v.ln_fold2) { need(n + ".ffn1.wl", wbytes(v, v.hidden, v.D)); need(n + ".ffn1.s", (size_t)v.hidden * 4); need(n + ".ffn1.bl", (size_t)v.hidden * 4); }
    if (v.stream_mlp) {
      if (v.D == 384) { need(n + ".embed.ws", (size_t)v.D * v.D * esz); need(n + ".qkv.ws", (size_t)3 * v.D * v.D * esz); }
      need(n + ".proj.ws", (size_t)v.D * v.D * esz); need(n + ".ffn.ws", (size_t)2 * v.D * v.hidden * esz); need(n + ".head.ws", (size_t)2 * v.D * v.hidden * esz);
    }
    if (v.gstream) {
      need(n + ".embed.ws", (size_t)v.D * v.D * esz); need(n + ".qkv.ws", (size_t)3 * v.D * v.D * esz);
      need(n + ".proj.ws", (size_t)v.D * v.D * esz); need(n + ".ffn.ws", (size_t)2 * v.D * v.hidden * esz); need(n + ".head.ws", (size_t)2 * v.D * v.hidden * esz);
    }
    if (v.chain) {
      CFEN_CHECK_ARG(v.ln_fold1 && v.ln_fold2, "net: %s: the persistent chain needs the LayerNorm-folded entries", n.c_str());
      need(n + ".embed.wf", (size_t)v.D * v.D * esz); need(n + ".qkv.wf", (size_t)3 * v.D * v.D * esz); need(n + ".proj.wf", (size_t)v.D * v.D * esz);
      need(n + ".ffn1.wf", (size_t)v.hidden * v.D * esz); need(n + ".ffn2.wf", (size_t)v.hidden * v.D * esz);
      need(n + ".head1.wf", (size_t)v.hidden * v.D * esz); need(n + ".head2.wf", (size_t)v.hidden * v.D * esz);
    }
    need(n + ".ln2.g", (size_t)v.D * 4); need(n + ".ln2.b", (size_t)v.D * 4);
    const char* wn = v.fused_mlp ? ".wk" : ".w";
    need(n + ".ffn1" + wn, wbytes(v, v.hidden, v.D)); need(n + ".ffn1.b", (size_t)v.hidden * 4);
    need(n + ".ffn2" + wn, wbytes(v, v.D, v.hidden)); need(n + ".ffn2.b", (size_t)v.D * 4);
    need(n + ".head1" + wn, wbytes(v, v.hidden, v.D)); need(n + ".head1.b", (size_t)v.hidden * 4);
    need(n + ".head2" + wn, wbytes(v, v.D, v.hidden)); need(n + ".head2.b", (size_t)v.D * 4);
  }

  // ---- convolution layers ----
  const int h = cfs ? nf : nf / 2;
  add_conv("head.0.0", 0, 5, 1, 2, 0, 1, 3, h, full);
  head5 = cfen_head5_supported(cfg.dtype, convs.at("head.0.0").Cout_pad, cs_of(h), full, full);
  if (head5) need("head.0.0.w5", (size_t)16 * 160 * esz);   // the 5x5 kernel with 8-byte pixels (k_head5.hip), beside the rows layout the A/B twin reads
  add_conv("head.0.1.body.0", 0, 3, 1, 1, 0, 1, h, h, full);
  add_conv("head.0.1.body.2", 0, 3, 1, 1, 0, 1, h, h, full);
  if (!cfs) add_conv("ds_conv_e01", 0, 3, 2, 1, 0, 1, h, nf, N);
  add_conv("ds_conv_e02", 0, 3, 2, 1, 0, 1, nf, 2 * nf, N / 2);
  add_conv("ds_conv_e03", 0, 3, 2, 1, 0, 1, 2 * nf, 4 * nf, N / 4);
  for (int l = 1; l <= 3; ++l) add_conv("lgcat_conv_e0" + std::to_string(l), 0, 1, 1, 0, 0, 2, nf << (l - 1), nf << (l - 1), N >> (l - 1));
  for (int b = 0; b < 3; ++b) {
    const std::string t(1, br[b]);
    for (int l = 1; l <= 3; ++l)
      add_conv("lgcat_conv_d0" + std::to_string(l) + t, 0, 1, 1, 0, 0, 2, nf << (l - 1), nf << (l - 1), N >> (l - 1));
    add_conv("us_conv_d03" + t, 1, 4, 2, 1, 0, 1, 4 * nf, 2 * nf, N / 2);
    add_conv("us_conv_d02" + t, 1, 4, 2, 1, 0, 1, 2 * nf, nf, N);
    if (!cfs) add_conv("us_conv_d01" + t, 1, 4, 2, 1, 0, 1, nf, h, 2 * N);
    if (b < 2 || crs) {   // crs:327-330: sk_conv_d03d / sk_conv_d02d read three maps
      add_conv("sk_conv_d03" + t, 0, 1, 1, 0, 0, b < 2 ? 2 : 3, 2 * nf, 2 * nf, N / 2);
      add_conv("sk_conv_d02" + t, 0, 1, 1, 0, 0, b < 2 ? 2 : 3, nf, nf, N);
    }
    const std::string T(1, (char)(br[b] - 32));
    add_conv("tail_" + T + ".conv3", 0, 3, 1, 1, 0, 1, h, h, full);
    add_conv("tail_" + T + ".conv7", 0, 7, 1, 3, 1, 1, h, b == 1 ? 1 : 3, full);
  }
  if (!crs) {
    need("cfsm2g_d03d.w", (size_t)4 * 2 * (2 * nf / 4) * (2 * nf) * 4);
    need("cfsm2g_d02d.w", (size_t)4 * 2 * (nf / 4) * nf * 4);
  }
  for (const Vit& v : vits)
    if (v.shrink) {     // v5:1101-1104: Conv2d 1x1 + ActNorm2d + ReLU on either side of the block
      add_conv(v.name + ".shrink", 0, 1, 1, 0, 0, 1, v.Cmap, v.Cmap / 4, v.mapH);
      add_conv(v.name + ".extend", 0, 1, 1, 0, 0, 1, v.Cmap / 4, v.Cmap, v.mapH);
    }

  // ---- workspace ----
  add_map("input", 3, full);
  add_map("head.conv5", h, full);
  add_map("head.res_mid", h, full);
  add_map("head", h, full);
  if (!cfs) add_map("ds_conv_e01", nf, N);
  for (int l = 1; l <= 3; ++l) {
    const std::string L = std::to_string(l);
    const int C = nf << (l - 1), E = N >> (l - 1);
    add_map("localvit_encoder_0" + L, C, E);
    add_map("globalvit_encoder_0" + L, C, E);
    add_map("lgcat_conv_e0" + L, C, E);
    if (l < 3) add_map("ds_conv_e0" + std::to_string(l + 1), 2 * C, E / 2);
  }
  for (int b = 0; b < 3; ++b) add_map(std::string("us_conv_d03") + br[b], 2 * nf, N / 2);   // back to back: one InstanceNorm over 3B images
  for (int b = 0; b < 3; ++b) {
    const std::string t(1, br[b]);
    for (int l = 3; l >= 1; --l) {
      const std::string L = std::to_string(l);
      const int C = nf << (l - 1), E = N >> (l - 1);
      add_map("localvit_decoder_0" + L + t, C, E);
      add_map("globalvit_decoder_0" + L + t, C, E);
      add_map("lgcat_conv_d0" + L + t, C, E);
      if (l < 3 && !(cfs && l == 1)) add_map("us_conv_d0" + L + t, C / 2, 2 * E);
      if (l > 1) add_map(b == 2 && !crs ? "cfsm2g_d0" + L + "d" : "sk_conv_d0" + L + t, C / 2, 2 * E);
    }
    add_map(std::string("tail_") + (char)(br[b] - 32) + ".mid", h, full);
  }
  for (const Vit& v : vits)
    if (v.shrink) {
      add_map(v.name + ".shrunk", v.Cmap / 4, v.mapH);   // conv_shrink output = the map the tokens are cut from
      add_map(v.name + ".tok", v.Cmap / 4, v.mapH);      // folded block output = conv_extend input
    }
  for (int k = 0; k < 6; ++k) {
    const bool g = k & 1;             // odd sets serve GViT lanes: far fewer tokens
    const size_t md = g ? max_md_g : max_md_l, mh = g ? max_mh_g : max_mh_l;
    Scratch& q = scr_set[k];
    q.x0 = alloc(md * esz); q.x1 = alloc(md * esz); q.yn = alloc(md * esz); q.att = alloc(md * esz);
    q.qkv = alloc(3 * md * esz);
    q.hid = alloc(mh * esz);
    q.small = alloc(g ? max_small * esz : 256);
    q.splitk = g ? alloc(SPLITK_BYTES) : 0;
    q.sync = g ? alloc(GV_SYNC_WORDS * sizeof(unsigned)) : 0;
  }
  for (int k = 0; k < 3; ++k) o_stats_set[k] = alloc(cfen_stats_workspace_bytes(3 * B, 128));   // [0] also serves the 3B-image InstanceNorm
  parallel = (cfg.reserved & 1) == 0;
  return CFEN_OK;
}

int cfen_net::run_conv_g(int ng, const ConvCall* cc, int act) {
  ConvDesc d[CFEN_MAX_GROUPS];
  if (!an_pending.empty() && !an_raw_pass) {
    bool any = false;
    for (int g = 0; g < ng; ++g) any = any || an_pending.count(cc[g].layer);
    if (any && ng > 1) {   // initialise member by member
      for (int g = 0; g < ng; ++g) TRY(run_conv_g(1, cc + g, act));
      return CFEN_OK;
    }
    if (any) {
      CFEN_CHECK_ARG(!cfen_recorder() && !profiling, "net: ActNorm '%s' is uninitialised; run one plain forward before capturing / profiling",
                     cc[0].layer.c_str());
      CFEN_CHECK_ARG(!cc[0].nchw_out, "net: ActNorm init on an NCHW output layer");
      const AnPending ap = an_pending.at(cc[0].layer);
      const ConvLayer& c = convs.at(cc[0].layer);
      const Buf& bo = bufs.at(cc[0].out);
      const std::string pending_layer = cc[0].layer;   // forgotten only once the statistics pass has been enqueued (see below)
      // 1. raw layer output x = conv + conv_bias (no activation, no residual) into the layer's own buffer
      Param keep_s = params.at(cc[0].layer + ".scale"), keep_t = params.at(cc[0].layer + ".shift");
      params[cc[0].layer + ".scale"].ptr = ap.ones;
      params[cc[0].layer + ".shift"].ptr = ap.conv_bias;
      ConvCall raw = cc[0];
      raw.res0.clear(); raw.res1.clear();
      an_raw_pass = true;                 // the raw pass is this same function: it must not start another initialisation of the layer
      int rc = run_conv_g(1, &raw, 0);
      an_raw_pass = false;
      params[cc[0].layer + ".scale"] = keep_s;
      params[cc[0].layer + ".shift"] = keep_t;
      if (rc) return rc;
      // 2. batch statistics -> epilogue table (written in place) + raw ActNorm parameters
      const void* xs = map_ptr(cc[0].out);
      int hw = bo.H * bo.W;
      if (ap.win > 0 && ap.win < bo.H) {
        // the reference module saw one window per call: statistics over the top-left window of every image, gathered into LViT scratch
        // (free here: the shrink conv runs before its block, the extend conv after the block's fold)
        unsigned char* tmp = (unsigned char*)at(scr_set[0].x0);
        const size_t row = (size_t)ap.win * bo.cs * esz, img = row * ap.win;
        for (int b = 0; b < cfg.batch; ++b)
          if (hipMemcpy2DAsync(tmp + b * img, row, (const unsigned char*)xs + (size_t)b * bo.H * bo.W * bo.cs * esz, (size_t)bo.W * bo.cs * esz, row,
                               ap.win, hipMemcpyDeviceToDevice, stream) != hipSuccess) {
            cfen_set_error("net: window copy for the ActNorm init of '%s' failed", cc[0].layer.c_str());
            return CFEN_ERR_HIP;
          }
        xs = tmp;
        hw = ap.win * ap.win;
      }
      TRY(cfen_actnorm_init_impl(cfg.dtype, xs, (float*)at(o_stats_set[2]), cfg.batch, hw, c.Cout, bo.cs, c.Cout_pad,
                                 ap.conv_bias, (float*)const_cast<void*>(keep_s.ptr), (float*)const_cast<void*>(keep_t.ptr), ap.an_out, stream));
      an_pending.erase(pending_layer);   // a failure above leaves the layer pending: the next forward retries instead of running it uninitialised
      // 3. fall through: the layer again, now with its real epilogue
    }
  }
  const ConvLayer& c0 = convs.at(cc[0].layer);
  double fl = 0.0;
  for (int g = 0; g < ng; ++g) {
    const ConvCall& q = cc[g];
    const ConvLayer& c = convs.at(q.layer);
    CFEN_CHECK_ARG(c.kind == c0.kind && c.k == c0.k && c.tile == c0.tile && c.tz == c0.tz && c.Cout_pad == c0.Cout_pad && c.Kpad == c0.Kpad,
                   "net: %s and %s cannot share a launch", cc[0].layer.c_str(), q.layer.c_str());
    const Buf& bi = bufs.at(q.in0);
    if (c.kind == 0)
      cfen_desc_conv(&d[g], cfg.batch, bi.H, bi.W, bi.cs, c.Cin, c.k, c.stride, c.pad, c.reflect, c.nsrc);
    else
      cfen_desc_convT4(&d[g], cfg.batch, bi.H, bi.W, bi.cs, c.Cin);
    d[g].src[0] = map_ptr(q.in0);
    d[g].src[1] = q.in1.empty() ? nullptr : map_ptr(q.in1);
    d[g].src[2] = q.in2.empty() ? nullptr : map_ptr(q.in2);
    if (q.up_low) { d[g].src[1] = q.up_low; d[g].up4 = 1; d[g].up_h = q.up_h; d[g].up_w = q.up_w; d[g].up_cs = q.up_cs; }
    CFEN_CHECK_ARG((c.nsrc >= 2) == !q.in1.empty() && (c.nsrc >= 3) == !q.in2.empty(), "net: %s reads %d maps", q.layer.c_str(), c.nsrc);
    d[g].weight = P(q.layer + (c.tz ? ".wz" : c.tile ? ".wr" : ".w")); d[g].Kpad = c.Kpad;
    d[g].scale = Pf(q.layer + ".scale"); d[g].shift = Pf(q.layer + ".shift");
    d[g].act = act;
    d[g].Cout = c.Cout; d[g].Cout_pad = c.Cout_pad;
    if (q.nchw_out) {
      d[g].out = q.nchw_out; d[g].out_nchw_f32 = (output_u8 && c.tz) ? 2 : 1; d[g].cs_out = c.Cout_pad;
      CFEN_CHECK_ARG(!output_f16, "net: fp16 outputs are written by the fused tail launch only (\"net.tail_fused\" = 2, stages not kept), %s runs on its own", q.layer.c_str());
      CFEN_CHECK_ARG(!output_u8 || c.tz, "net: uint8 outputs need the Toeplitz 7x7 tail kernel, %s does not run on it", q.layer.c_str());
    } else {
      const Buf& bo = bufs.at(q.out);
      CFEN_CHECK_ARG(bo.H == d[g].Hout && bo.W == d[g].Wout, "net: %s output geometry mismatch", q.layer.c_str());
      d[g].out = map_ptr(q.out); d[g].cs_out = bo.cs; d[g].cs_res = bo.cs;
      d[g].res[0] = q.res0.empty() ? nullptr : map_ptr(q.res0);
      d[g].res[1] = q.res1.empty() ? nullptr : map_ptr(q.res1);
    }
    const double e = (double)c.out_edge;
    fl += cfg.batch * (c.kind == 1 ? 2.0 * c.Cin_real * c.Cout * 16.0 * (e / 2) * (e / 2)
                                   : 2.0 * c.Cout * (double)c.Cin_real * c.nsrc * c.k * c.k * e * e);
  }
  label = cc[0].layer + (ng > 1 ? " (x" + std::to_string(ng) + ")" : "");
  if (c0.tz)
    TRYP(K_CONV, fl, cfen_conv7_tz_impl_g(cfg.dtype, ng, d, stream));
  else if (c0.tile && c0.kind == 1)
    TRYP(K_CONV, fl, cfen_convT_tile_impl_g(cfg.dtype, ng, d, stream));
  else if (c0.tile)
    TRYP(K_CONV, fl, cfen_conv_tile_impl_g(cfg.dtype, ng, d, c0.k, stream));
  else
    TRYP(K_CONV, fl, cfen_conv_impl_g(cfg.dtype, ng, d, stream));
  return CFEN_OK;
}

// One LViT / GViT instance: reference v3:1136-1189 / 1272-1325 (+ TransformerEncoderLayer 1382-1390).
int cfen_net::run_vit_g(int ng, const VitCall* vc, int scr0) {
  const int dt = cfg.dtype, B = cfg.batch;
  const Vit& v = *vc[0].v;
  const Buf& bi = bufs.at(vc[0].in);
  const Buf& bo = bufs.at(vc[0].out);
  const int nwin = (v.mapH / v.ws) * (v.mapH / v.ws);
  const int M = B * nwin * v.S;
  blk_kind = v.global ? 1 : 2;
  struct Reset { int& k; ~Reset() { k = 0; } } reset_{blk_kind};
  CFEN_CHECK_ARG(v.global == (bool)(scr0 & 1) && scr0 + 2 * (ng - 1) < 6, "net: %s enqueued on the wrong scratch set", v.name.c_str());
  // per-member operands: scratch set scr0 + 2g, parameters by instance name
  void *X0[3], *X1[3], *YN[3], *QKV[3], *ATT[3], *HID[3], *SM[3], *OUT[3];
  const void *IN[3], *cX0[3], *cX1[3], *cYN[3], *cQKV[3], *cSM[3];
  std::string nm[3];
  for (int g = 0; g < ng; ++g) {
    const Vit& w = *vc[g].v;
    CFEN_CHECK_ARG(w.global == v.global && w.D == v.D && w.Dn == v.Dn && w.Da == v.Da && w.hidden == v.hidden && w.S == v.S && w.mapH == v.mapH && w.heads == v.heads &&
                   w.fused_mlp == v.fused_mlp && w.fused_front == v.fused_front && bufs.at(vc[g].in).cs == bi.cs && bufs.at(vc[g].out).cs == bo.cs,
                   "net: %s and %s cannot share launches", v.name.c_str(), w.name.c_str());
    const Scratch& q = scr_set[scr0 + 2 * g];
    X0[g] = at(q.x0); X1[g] = at(q.x1); YN[g] = at(q.yn); QKV[g] = at(q.qkv); ATT[g] = at(q.att); HID[g] = at(q.hid);
    SM[g] = v.global ? at(q.small) : nullptr;
    IN[g] = map_ptr(vc[g].in); OUT[g] = map_ptr(vc[g].out);
    cX0[g] = X0[g]; cX1[g] = X1[g]; cYN[g] = YN[g]; cQKV[g] = QKV[g]; cSM[g] = SM[g];
    nm[g] = w.name;
  }
  const void* cHIDp[3] = {HID[0], HID[1], HID[2]};
  float* SK[3] = {nullptr, nullptr, nullptr};   // split-K scratch of each member (GViT: few tokens against 0.5 GB of weights per forward)
  if (v.global)
    for (int g = 0; g < ng; ++g) SK[g] = (float*)at(scr_set[scr0 + 2 * g].splitk);
  const double Md = (double)M * ng, D = v.Dn, Hd = v.hidden;   // algorithmic flops count the real embedding dim
  auto step = [&](const char* what) { if (profiling) label = nm[0] + (ng > 1 ? " (x" + std::to_string(ng) + ")" : "") + ":" + what; };
  for (int i = 0; i < cfen_tune_extra_launches(); ++i) {
    step("extra_launch");
    TRY(cfen_occupy_impl(-1, 1, 1, 0, base, ws_bytes, X0[0], stream));
  }
  if (v.global && cfen_tune_gvit_dummy_wgs() != 0 && (cfen_tune_gvit_dummy_levels() == 0 || (!v.gstream && (cfen_tune_gvit_dummy_levels() == 1 || (cfen_tune_gvit_dummy_levels() == 2) == (v.D >= 1536))))) {   // levels: 1 = 2 and 3, 2 = level 3 only, 3 = level 2 only   // what-if probe: the whole block replaced by a launch that holds CUs (outputs invalid)
    step("dummy");
    return cfen_occupy_impl(cfen_tune_gvit_dummy_wgs(), ng, cfen_tune_gvit_dummy_us(), cfen_tune_gvit_dummy_stream(), base, ws_bytes, SK[0], stream);
  }
  // Y = act(X W^T + bias) + R + P for every member; operand arrays are indexed by member
  auto gemm = [&](const void* const* X, const char* wname, const char* bname, void* const* R, const char* pname, void* const* Y, int N, int K,
                  int relu, const CfenTokGather* tg) -> int {
    CfenGemmPtrs gp[3];
    for (int g = 0; g < ng; ++g)
      gp[g] = CfenGemmPtrs{tg ? nullptr : X[g], P(nm[g] + wname), bname ? Pf(nm[g] + bname) : nullptr, R ? R[g] : nullptr,
                           pname ? P(nm[g] + pname) : nullptr, Y[g], tg ? IN[g] : nullptr, nullptr, v.global && wtile};
    return cfen_gemm_impl_g(dt, ng, gp, K, K, N, v.S, N, M, N, K, relu, tg, stream, v.global ? SK : nullptr, v.global ? SPLITK_BYTES : 0);
  };
  // algorithmic bytes of a token GEMM launch: weights once + tokens in + tokens out (what a weight-streaming GEMM is priced against)
  auto gemm_bytes = [&](int N, int K) { prof_bytes = (double)ng * ((double)N * K + (double)M * K + (double)M * N) * esz; };
  // ... of a fused token-block launch (round 6: priced for the MFMA-bound kernels too, so that counter traffic / algorithmic bytes can be read from the bench line):
  // `rows` token-row-sized tensors in and out (M x D elements each: map or token reads, X1 / QKV / ATT / map writes) + `welems` weight elements read once
  auto block_bytes = [&](double rows, double welems) { prof_bytes = (double)ng * (rows * (double)M * v.D + welems) * esz; };
  const double DD = (double)v.D * v.D, DH = (double)v.D * v.hidden;
  // Y = act(LN(X) W0^T + b0) with the LayerNorm folded: parameters `lname`.wl / .s / .bl (packing.ln_folded)
  auto gemm_ln = [&](const void* const* X, const std::string& lname, void* const* Y, int N, int K, int relu) -> int {
    CfenGemmPtrs gp[3];
    for (int g = 0; g < ng; ++g)
      gp[g] = CfenGemmPtrs{X[g], P(nm[g] + lname + ".wl"), Pf(nm[g] + lname + ".bl"), nullptr, nullptr, Y[g], nullptr, Pf(nm[g] + lname + ".s"),
                           v.global && wtile};
    return cfen_gemm_impl_g(dt, ng, gp, K, K, N, v.S, N, M, N, K, relu, nullptr, stream, v.global ? SK : nullptr, v.global ? SPLITK_BYTES : 0);
  };
  if (v.gstream && (cfen_tune_gvit_stream() == 2 || (cfen_tune_gvit_stream() == 1 && !parallel)) && cfen_front3_supported(dt, v.D, (long long)M)) {
    // GViT block of embedding dim 384 (level 1) on the LViT-3 stream kernels: 4 x 4 pooled MAP -> k_front3 (patch gather + linear_encoding + residual +
    // position + LN1 + in_proj, qkv row-major) -> attention -> k_mlp3 (out_proj + LN2 + FFN + mlp_head + fold into the pooled-size map) -> x4
    // bilinear: 5 launches instead of 10, and the two stream launches are 16 whole-CU workgroups per block -- 2-3x the latency of the GEMM chain and
    // a fraction of its CU-time, which is what counts with several forwards in flight
    const void* PM[3];
    for (int g = 0; g < ng; ++g) PM[g] = X0[g];          // the token scratch holds the pooled map: B x mapH x mapH x C = M x D elements
    step("pool4");
    TRYP(K_TOKEN, 0, cfen_pool4_impl_g(dt, ng, IN, X0, B, v.mapH, v.mapH, v.C, bi.cs, v.C, stream));
    CfenEmbedQkvArgs e[3];
    for (int g = 0; g < ng; ++g)
      e[g] = CfenEmbedQkvArgs{PM[g], B, v.mapH, v.mapH, v.C, v.C, v.ws, v.p, P(nm[g] + ".embed.ws"), Pf(nm[g] + ".embed.b"), P(nm[g] + ".pos"),
                              Pf(nm[g] + ".ln1.g"), Pf(nm[g] + ".ln1.b"), P(nm[g] + ".qkv.ws"), X1[g], QKV[g], M, v.D, 1e-5f, 0};
    step("front_stream");
    block_bytes(5, 4 * DD);     // pixels in, X1 + q / k / v out; W_e + W_qkv
    TRYP(K_GEMM, 8 * Md * D * D, cfen_front3_impl_g(dt, ng, e, stream));
    step("attention");
    block_bytes(4, 0);   // q, k, v in; attention output out
    TRYP(K_ATTN, 4 * Md * v.S * D, cfen_attention_impl_g(dt, ng, cQKV, ATT, B * nwin, v.S, v.heads, v.dh, stream));
    Mlp3Args m[3];
    for (int g = 0; g < ng; ++g) {
      const std::string& n = nm[g];
      m[g] = Mlp3Args{};
      m[g].X = X1[g]; m[g].A = ATT[g]; m[g].Wp = P(n + ".proj.ws"); m[g].Y = nullptr; m[g].fmap = SM[g];
      m[g].ln_g = Pf(n + ".ln2.g"); m[g].ln_b = Pf(n + ".ln2.b");
      m[g].Wa = P(n + ".ffn.ws"); m[g].b1a = Pf(n + ".ffn1.b"); m[g].b2a = Pf(n + ".ffn2.b");
      m[g].Wb = P(n + ".head.ws"); m[g].b1b = Pf(n + ".head1.b"); m[g].b2b = Pf(n + ".head2.b");
      m[g].M = M; m[g].D = v.D; m[g].H = v.hidden; m[g].eps = 1e-5f;
      m[g].mapH = v.mapH; m[g].mapW = v.mapH; m[g].C = v.C; m[g].cs = v.C; m[g].ws = v.ws; m[g].p = v.p;
    }
    step("proj_mlp_stream");
    block_bytes(3, DD + 4 * DH);   // X1 + attention output in, map out; W_p + the four MLP matrices
    TRYP(K_MLP, 8 * Md * D * Hd + 2 * Md * D * D, cfen_mlp3_impl_g(dt, ng, m, stream));
    if (gv_skip_up) return CFEN_OK;
    step("upsample4");
    TRYP(K_TOKEN, 0, cfen_upsample4_impl_g(dt, ng, cSM, OUT, B, v.mapH, v.mapH, v.C, v.C, bo.cs, stream));
    return CFEN_OK;
  }
  if (v.chain && (cfen_tune_gvit_chain() == 1 || cfen_tune_gvit_chain() == 4 || cfen_tune_gvit_chain() == 5 || (cfen_tune_gvit_chain() == 2 && ng > 1) || (cfen_tune_gvit_chain() == 3 && ng == 1))) {
    // GViT block: pooled patch tokens -> [embed -> qkv] -> attention -> [proj -> ffn1 -> ffn2 -> head1 -> head2 + fold] -> x4 bilinear; the two
    // bracketed runs are ONE persistent launch each (k_gvit.hip): a team of workgroups per block keeps its CUs over the whole run
    const int team = std::max(1, std::min(cfen_tune_gvit_team(), 256 / (ng * std::max(1, cfen_tune_gvit_max_concurrent()))));   // every team of every forward in flight must be RESIDENT at once (grid barrier)
    const bool per_gemm = cfen_tune_gvit_chain() >= 4;   // 4: every GEMM its own launch of the chain kernel, never split over K: no grid barrier, no split-K seam;
                                                         // 5 (round 5 A/B): as 4 with K split so that a launch has ~a workgroup per CU (in-launch split-K seam, no grid barrier)
    const int split_team = cfen_tune_gvit_chain() == 5 ? 256 / ng : team;
    auto nsp = [&](int N, int K) {   // K slices so that the phase has work for most of the team (>= 4 K-steps of 64 per slice)
      int n = 1;
      if (cfen_tune_gvit_chain() == 4) return n;
      const int units = ((M + 127) / 128) * (N / 128);
      while (units * n * 2 <= split_team && n < 8 && (K / 64) % (2 * n) == 0 && K / 64 / (2 * n) >= 4) n *= 2;
      return n;
    };
    auto sync_of = [&](int g, CfenChainArgs& c) {
      unsigned* w = (unsigned*)at(scr_set[scr0 + 2 * g].sync);
      CFEN_CHECK_ARG(gv_launch < GV_ERR_WORD, "net: too many persistent-chain launches in one forward");
      c.bar = w + gv_launch; c.err = w + GV_ERR_WORD;
      c.cnt = (unsigned*)SK[g]; c.ncnt = CFEN_SPLITK_COUNTERS;
      c.part = SK[g] + CFEN_SPLITK_COUNTERS * sizeof(unsigned) / sizeof(float); c.part_bytes = SPLITK_BYTES - CFEN_SPLITK_COUNTERS * sizeof(unsigned);
      return CFEN_OK;
    };
    step("patchify");
    TRYP(K_TOKEN, 0, cfen_patchify_impl_g(dt, ng, IN, X0, B, v.mapH, v.mapH, v.C, bi.cs, v.ws, v.p, 4, 0, stream));
    CfenChainArgs ca[3];
    for (int g = 0; g < ng; ++g) {
      const std::string& n = nm[g];
      CfenChainArgs& c = ca[g];
      c = CfenChainArgs{};
      c.nph = 2; c.M = M;
      //                      X       W                  bias                 lnf_s              R       P             Y       ldx   ldr   ldy     period N       K    relu nsplit fold
      c.ph[0] = CfenChainPhase{X0[g], P(n + ".embed.wf"), Pf(n + ".embed.b"), nullptr, X0[g], P(n + ".pos"), X1[g], v.D, v.D, v.D, v.S, v.D, v.D, 0, nsp(v.D, v.D), 0};
      c.ph[1] = CfenChainPhase{X1[g], P(n + ".qkv.wf"), Pf(n + ".qkv.bl"), Pf(n + ".qkv.s"), nullptr, nullptr, QKV[g], v.D, 0, 3 * v.D, 0, 3 * v.D, v.D, 0, 1, 0};
      TRY(sync_of(g, c));
    }
    ++gv_launch;
    auto run_chain = [&](const char* what, double fl) -> int {
      if (!per_gemm) {
        step(what);
        TRYP(K_GEMM, fl, cfen_gvit_chain_impl_g(dt, ng, ca, team, stream));
        return CFEN_OK;
      }
      const int nph = ca[0].nph;
      for (int p = 0; p < nph; ++p) {
        CfenChainArgs one[3];
        for (int g = 0; g < ng; ++g) {
          one[g] = ca[g];
          one[g].ph[0] = ca[g].ph[p];
          one[g].nph = 1;
        }
        const int units = ((M + 127) / 128) * (one[0].ph[0].N / 128) * std::max(1, one[0].ph[0].nsplit);
        const std::string lab = std::string(what) + "." + std::to_string(p);
        step(lab.c_str());
        // (no grid barrier in a one-phase launch: the residency cap of the persistent chains does not apply)
        TRYP(K_GEMM, fl / nph, cfen_gvit_chain_impl_g(dt, ng, one, std::min(256 / ng, units), stream));
      }
      return CFEN_OK;
    };
    TRY(run_chain("chain_embed_qkv", 8 * Md * D * D));
    step("attention");
    block_bytes(4, 0);   // q, k, v in; attention output out
    TRYP(K_ATTN, 4 * Md * v.S * D, cfen_attention_impl_g(dt, ng, cQKV, ATT, B * nwin, v.S, v.heads, v.dh, stream));
    for (int g = 0; g < ng; ++g) {
      const std::string& n = nm[g];
      CfenChainArgs& c = ca[g];
      c = CfenChainArgs{};
      c.nph = 5; c.M = M;
      c.fH = v.mapH; c.fW = v.mapH; c.fcs = v.C; c.fC = v.C; c.fp = v.p;
      c.ph[0] = CfenChainPhase{ATT[g], P(n + ".proj.wf"), nullptr, nullptr, X1[g], nullptr, X1[g], v.D, v.D, v.D, 0, v.D, v.D, 0, nsp(v.D, v.D), 0};
      c.ph[1] = CfenChainPhase{X1[g], P(n + ".ffn1.wf"), Pf(n + ".ffn1.bl"), Pf(n + ".ffn1.s"), nullptr, nullptr, HID[g], v.D, 0, v.hidden, 0, v.hidden, v.D, 1, 1, 0};
      c.ph[2] = CfenChainPhase{HID[g], P(n + ".ffn2.wf"), Pf(n + ".ffn2.b"), nullptr, X1[g], nullptr, X1[g], v.hidden, v.D, v.D, 0, v.D, v.hidden, 0, nsp(v.D, v.hidden), 0};
      c.ph[3] = CfenChainPhase{X1[g], P(n + ".head1.wf"), Pf(n + ".head1.b"), nullptr, nullptr, nullptr, HID[g], v.D, 0, v.hidden, 0, v.hidden, v.D, 1, 1, 0};
      c.ph[4] = CfenChainPhase{HID[g], P(n + ".head2.wf"), Pf(n + ".head2.b"), nullptr, X1[g], nullptr, SM[g], v.hidden, v.D, v.D, 0, v.D, v.hidden, 0, nsp(v.D, v.hidden), 1};
      TRY(sync_of(g, c));
    }
    ++gv_launch;
    TRY(run_chain("chain_proj_mlp_head", 2 * Md * D * D + 8 * Md * D * Hd));
    if (gv_skip_up) return CFEN_OK;
    step("upsample4");
    TRYP(K_TOKEN, 0, cfen_upsample4_impl_g(dt, ng, cSM, OUT, B, v.mapH, v.mapH, v.C, v.C, bo.cs, stream));
    return CFEN_OK;
  }
  if (v.fused_window && cfen_tune_lvit_window()) {
    // LViT level 1: one workgroup per window, embed -> attention -> MLP -> fold with q / k / v / attention output on chip (k_lvit.hip)
    LvitArgs w[3];
    for (int g = 0; g < ng; ++g) {
      const std::string& n = nm[g];
      w[g] = LvitArgs{IN[g], OUT[g], B, v.mapH, v.mapH, v.C, bi.cs, bo.cs, v.ws, v.p, P(n + ".lw.ws"), Pf(n + ".embed.b"), P(n + ".pos"),
                      Pf(n + ".ln1.g"), Pf(n + ".ln1.b"), Pf(n + ".ln2.g"), Pf(n + ".ln2.b"), Pf(n + ".ffn1.b"), Pf(n + ".ffn2.b"),
                      Pf(n + ".head1.b"), Pf(n + ".head2.b"), v.hidden, 1e-5f, 1.4426950408889634f / sqrtf((float)(v.D / v.heads))};
    }
    step("window_block_fused");
    block_bytes(2, 5 * DD + 4 * DH);   // map in, map out; W_e, W_qkv, W_p + the four MLP matrices
    TRYP(K_MLP, 8 * Md * D * D + 4 * Md * v.S * D + 8 * Md * D * Hd + 2 * Md * D * D, cfen_lvit_window_impl_g(dt, ng, w, stream));
    return CFEN_OK;
  }
  const float* lg[3];
  const float* lb[3];
  bool head_major = false;   // the fused front half writes qkv per (window, head) for k_attention_hm
  const long long Mll = M;
  if (v.stream_mlp && cfen_front3_supported(dt, v.D, Mll) && cfen_attention_hm_supported(dt, v.S, v.D / v.heads) &&
      (cfen_tune_stream_front() >= 2 || (cfen_tune_stream_front() == 1 && (long long)ng * M >= 128LL * 128))) {   // >= 128 workgroups of 128 tokens
    // LViT level 3: gather + linear_encoding + residual + position + LN1 + qkv in one launch on row-tile weight streams (k_stream.hip)
    CfenEmbedQkvArgs e[3];
    head_major = true;
    for (int g = 0; g < ng; ++g)
      e[g] = CfenEmbedQkvArgs{IN[g], B, v.mapH, v.mapH, v.C, bi.cs, v.ws, v.p, P(nm[g] + ".embed.ws"), Pf(nm[g] + ".embed.b"), P(nm[g] + ".pos"),
                              Pf(nm[g] + ".ln1.g"), Pf(nm[g] + ".ln1.b"), P(nm[g] + ".qkv.ws"), X1[g], QKV[g], M, v.D, 1e-5f, v.heads};
    step("front_stream");
    block_bytes(5, 4 * DD);     // pixels in, X1 + q / k / v out; W_e + W_qkv
    TRYP(K_GEMM, 8 * Md * D * D, cfen_front3_impl_g(dt, ng, e, stream));
  } else if (v.fused_front && v.D <= cfen_tune_fused_front_max_dim()) {
    // LViT levels 1-2: gather + linear_encoding + residual + position + LN1 + qkv in one launch, x -> X1, QKV (k_embed.hip)
    CfenEmbedQkvArgs e[3];
    head_major = cfen_tune_attn_head_major() && cfen_attention_hm_supported(dt, v.S, v.D / v.heads);
    for (int g = 0; g < ng; ++g)
      e[g] = CfenEmbedQkvArgs{IN[g], B, v.mapH, v.mapH, v.C, bi.cs, v.ws, v.p, P(nm[g] + ".embed.wk"), Pf(nm[g] + ".embed.b"), P(nm[g] + ".pos"),
                              Pf(nm[g] + ".ln1.g"), Pf(nm[g] + ".ln1.b"), P(nm[g] + ".qkv.wk"), X1[g], QKV[g], M, v.D, 1e-5f,
                              head_major ? v.heads : 0};
    step("embed_ln_qkv");
    block_bytes(5, 4 * DD);
    TRYP(K_GEMM, 8 * Md * D * D, cfen_embed_qkv_impl_g(dt, ng, e, stream));
  } else {
    if (v.global || !cfen_tune_embed_gather()) {
      step("patchify");
      TRYP(K_TOKEN, 0, cfen_patchify_impl_g(dt, ng, IN, X0, B, v.mapH, v.mapH, v.C, bi.cs, v.ws, v.p, v.global ? 4 : 1, 0, stream));
      // x = linear_encoding(x) + x + pos                                      (v3:1143,1166)
      step("embed");
      gemm_bytes(v.D, v.D);
      TRYP(K_GEMM, 2 * Md * D * D, gemm(cX0, ".embed.w", ".embed.b", X0, ".pos", X1, v.D, v.D, 0, nullptr));
    } else {
      // LViT: the window / patch gather rides on the embedding GEMM's loader, no token buffer is written
      step("embed");
      CfenTokGather tg{nullptr, B, v.mapH, v.mapH, v.C, bi.cs, v.ws, v.p};
      TRYP(K_GEMM, 2 * Md * D * D, gemm(nullptr, ".embed.w", ".embed.b", nullptr, ".pos", X1, v.D, v.D, 0, &tg));
    }
    // src = src + out_proj(MHA(LN1(src)))                                      (v3:1383-1386)
    if (v.ln_fold1 && cfen_tune_ln_fold()) {
      step("ln1_qkv");
      gemm_bytes(3 * v.Da, v.D);
      TRYP(K_GEMM, 6 * Md * D * D, gemm_ln(cX1, ".qkv", QKV, 3 * v.Da, v.D, 0));
    } else {
      step("ln1");
      for (int g = 0; g < ng; ++g) { lg[g] = Pf(nm[g] + ".ln1.g"); lb[g] = Pf(nm[g] + ".ln1.b"); }
      TRYP(K_LNORM, 0, cfen_layernorm_impl_g(dt, ng, cX1, YN, lg, lb, M, v.D, 1e-5f, stream, v.Dn));
      step("qkv");
      TRYP(K_GEMM, 6 * Md * D * D, gemm(cYN, ".qkv.w", nullptr, nullptr, nullptr, QKV, 3 * v.Da, v.D, 0, nullptr));
    }
  }
  step("attention");
  block_bytes(4, 0);   // q, k, v in; attention output out
  if (head_major)
    TRYP(K_ATTN, 4 * Md * v.S * D, cfen_attention_hm_impl_g(dt, ng, cQKV, ATT, B * nwin, v.S, v.heads, v.D / v.heads, stream));
  else
    TRYP(K_ATTN, 4 * Md * v.S * D, cfen_attention_impl_g(dt, ng, cQKV, ATT, B * nwin, v.S, v.heads, v.dh, stream));
  const void* cATT[3] = {ATT[0], ATT[1], ATT[2]};
  // D = 384: launches of >= 128 workgroups of 128 tokens; D = 192 (LViT level 2): always -- k_mlp3<12, 3> is spill-free and 192-token workgroups
  // make the grouped decoder launch exactly two rounds (154-162 us against k_mlp2's 181-204, tools/bench_mlp3.py)
  const bool stream_mlp = v.stream_mlp && (v.D == 192 ? cfen_tune_stream_mlp192() != 0
                                                      : (cfen_tune_stream_mlp() >= 2 || (cfen_tune_stream_mlp() == 1 && (long long)ng * M >= 128LL * 128)));
  if (stream_mlp) {
    // LViT level 3: out_proj + residual + LN2 + FFN + mlp_head + fold in one launch on fragment-stream weights (k_stream.hip)
    Mlp3Args m[3];
    for (int g = 0; g < ng; ++g) {
      const std::string& n = nm[g];
      m[g] = Mlp3Args{};
      m[g].X = X1[g]; m[g].A = ATT[g]; m[g].Wp = P(n + ".proj.ws"); m[g].Y = nullptr; m[g].fmap = OUT[g];
      m[g].ln_g = Pf(n + ".ln2.g"); m[g].ln_b = Pf(n + ".ln2.b");
      m[g].Wa = P(n + ".ffn.ws"); m[g].b1a = Pf(n + ".ffn1.b"); m[g].b2a = Pf(n + ".ffn2.b");
      m[g].Wb = P(n + ".head.ws"); m[g].b1b = Pf(n + ".head1.b"); m[g].b2b = Pf(n + ".head2.b");
      m[g].M = M; m[g].D = v.D; m[g].H = v.hidden; m[g].eps = 1e-5f;
      m[g].mapH = v.mapH; m[g].mapW = v.mapH; m[g].C = v.C; m[g].cs = bo.cs; m[g].ws = v.ws; m[g].p = v.p;
    }
    step("proj_mlp_stream");
    block_bytes(3, DD + 4 * DH);   // X1 + attention output in, map out; W_p + the four MLP matrices
    TRYP(K_MLP, 8 * Md * D * Hd + 2 * Md * D * D, cfen_mlp3_impl_g(dt, ng, m, stream));
    return CFEN_OK;
  }
  if (!v.fused_mlp) {
    step("proj");
    gemm_bytes(v.D, v.Da);
    TRYP(K_GEMM, 2 * Md * D * D, gemm(cATT, ".proj.w", nullptr, X1, nullptr, X1, v.D, v.Da, 0, nullptr));
  }
  if (v.fused_mlp) {
    // LN2 + FFN + residual + mlp_head + residual + fold, hidden activations never leave registers (k_mlp.hip)
    MlpArgs m[3];
    for (int g = 0; g < ng; ++g) {
      const std::string& n = nm[g];
      m[g] = MlpArgs{};
      m[g].X = X1[g]; m[g].Y = nullptr; m[g].fmap = v.global ? SM[g] : OUT[g];
      m[g].A = ATT[g]; m[g].Wp = P(n + ".proj.w");   // out_proj + residual ride on the kernel's token load
      m[g].ln_g = Pf(n + ".ln2.g"); m[g].ln_b = Pf(n + ".ln2.b");
      m[g].W1a = P(n + ".ffn1.wk"); m[g].b1a = Pf(n + ".ffn1.b"); m[g].W2a = P(n + ".ffn2.wk"); m[g].b2a = Pf(n + ".ffn2.b");
      m[g].W1b = P(n + ".head1.wk"); m[g].b1b = Pf(n + ".head1.b"); m[g].W2b = P(n + ".head2.wk"); m[g].b2b = Pf(n + ".head2.b");
      m[g].M = M; m[g].D = v.D; m[g].H = v.hidden; m[g].eps = 1e-5f;
      m[g].mapH = v.mapH; m[g].mapW = v.mapH; m[g].C = v.C; m[g].cs = v.global ? v.C : bo.cs; m[g].ws = v.ws; m[g].p = v.p;
    }
    step("proj_mlp_fused");
    block_bytes(3, DD + 4 * DH);
    TRYP(K_MLP, 8 * Md * D * Hd + 2 * Md * D * D, cfen_mlp_impl_g(dt, ng, m, stream));
  } else {
    // src = src + linear2(relu(linear1(LN2(src))))                             (v3:1387-1389)
    const void* const* cHID = cHIDp;
    if (v.ln_fold2 && cfen_tune_ln_fold()) {
      step("ln2_ffn1");
      gemm_bytes(v.hidden, v.D);
      TRYP(K_GEMM, 2 * Md * D * Hd, gemm_ln(cX1, ".ffn1", HID, v.hidden, v.D, 1));
    } else {
      step("ln2");
      for (int g = 0; g < ng; ++g) { lg[g] = Pf(nm[g] + ".ln2.g"); lb[g] = Pf(nm[g] + ".ln2.b"); }
      TRYP(K_LNORM, 0, cfen_layernorm_impl_g(dt, ng, cX1, YN, lg, lb, M, v.D, 1e-5f, stream, v.Dn));
      step("ffn1");
      TRYP(K_GEMM, 2 * Md * D * Hd, gemm(cYN, ".ffn1.w", ".ffn1.b", nullptr, nullptr, HID, v.hidden, v.D, 1, nullptr));
    }
    step("ffn2");
    gemm_bytes(v.D, v.hidden);
    TRYP(K_GEMM, 2 * Md * D * Hd, gemm(cHID, ".ffn2.w", ".ffn2.b", X1, nullptr, X1, v.D, v.hidden, 0, nullptr));
    // x = mlp_head(x) + x                                                      (v3:1173)
    step("head1");
    gemm_bytes(v.hidden, v.D);
    TRYP(K_GEMM, 2 * Md * D * Hd, gemm(cX1, ".head1.w", ".head1.b", nullptr, nullptr, HID, v.hidden, v.D, 1, nullptr));
    const void* dst[3];
    for (int g = 0; g < ng; ++g) dst[g] = v.global ? SM[g] : OUT[g];
    if (cfen_tune_fold_in_gemm()) {
      // x = mlp_head(x) + x, folded: the GEMM's epilogue writes feature (i, j, c) of token m to its pixel of the map   (v3:1173, 1186)
      step("head2_fold");
      gemm_bytes(v.D, v.hidden);
      CfenGemmPtrs gp[3];
      for (int g = 0; g < ng; ++g)
        gp[g] = CfenGemmPtrs{HID[g], P(nm[g] + ".head2.w"), Pf(nm[g] + ".head2.b"), X1[g], nullptr, X0[g], nullptr, nullptr, v.global && wtile,
                             const_cast<void*>(dst[g])};
      const CfenTokGather yg{nullptr, B, v.mapH, v.mapH, v.C, v.global ? v.C : bo.cs, v.ws, v.p};
      TRYP(K_GEMM, 2 * Md * D * Hd, cfen_gemm_impl_g(dt, ng, gp, v.hidden, v.hidden, v.D, v.S, v.D, M, v.D, v.hidden, 0, nullptr, stream,
                                                    v.global ? SK : nullptr, v.global ? SPLITK_BYTES : 0, &yg));
    } else {
      step("head2");
      TRYP(K_GEMM, 2 * Md * D * Hd, gemm(cHID, ".head2.w", ".head2.b", X1, nullptr, X0, v.D, v.hidden, 0, nullptr));
      step("fold");
      TRYP(K_TOKEN, 0, cfen_patchify_impl_g(dt, ng, dst, X0, B, v.mapH, v.mapH, v.C, v.global ? v.C : bo.cs, v.ws, v.p, 1, 1, stream));
    }
  }
  if (v.global && gv_skip_up) return CFEN_OK;
  step("upsample4");
  if (v.global) TRYP(K_TOKEN, 0, cfen_upsample4_impl_g(dt, ng, cSM, OUT, B, v.mapH, v.mapH, v.C, v.C, bo.cs, stream));
  return CFEN_OK;
}

// LViT || GViT -> 1x1 fuse conv over their concat -> ActNorm -> ReLU -> + level input   (v3:403-488 ...)
// for `ng` same-level blocks at once (tags "e" = the encoder block, "rsd" = the three decoders).  GViT (few tokens, huge
// weights: latency / weight-bandwidth bound) is enqueued on the side stream `sg` and overlaps LViT (many tokens).
int cfen_net::run_level_g(int ng, const char* tags, int l, const std::string* in, const char* extra_res, const std::string* out,
                          hipStream_t sm, hipStream_t sg) {
  const std::string L = std::to_string(l);
  VitCall lv[3], gv[3];
  ConvCall fuse[3];
  for (int g = 0; g < ng; ++g) {
    const bool enc = tags[g] == 'e';
    const std::string ln = enc ? "localvit_encoder_0" + L : "localvit_decoder_0" + L + tags[g];
    const std::string gn = enc ? "globalvit_encoder_0" + L : "globalvit_decoder_0" + L + tags[g];
    lv[g] = VitCall{nullptr, in[g], ln};
    gv[g] = VitCall{nullptr, in[g], gn};
    for (const Vit& v : vits) {
      if (v.name == ln) lv[g].v = &v;
      if (v.name == gn) gv[g].v = &v;
    }
    fuse[g] = ConvCall{out[g], ln, gn, in[g], extra_res ? extra_res : "", out[g], nullptr};
  }
  // globalvit_encoder_02 has its own hidden size (v3:200), so blocks of one level share launches only within the decoders
  // x4 bilinear of the GViT result inside the fuse conv (k_conv UP): the block leaves its low-resolution map in the scratch set, no k_upsample4 launch
  // and no full-resolution copy.  Not while an ActNorm of the level is still uninitialised (its raw pass reads the stored map) or stages are asked for.
  {
    const Vit& g0 = *gv[0].v;
    const ConvLayer& fc = convs.at(out[0]);
    const Buf& bo = bufs.at(out[0]);
    const bool up = cfen_tune_up_fused() && !cfen_tune_keep_stages() && an_pending.empty() && cfg.dtype == CFEN_F16 && !g0.shrink && !lv[0].v->shrink &&
                    fc.kind == 0 && fc.k == 1 && fc.nsrc == 2 && !fc.tile && fc.Cin == g0.C && bo.H == 4 * g0.mapH && bo.W == 4 * g0.mapH &&
                    cfen_conv_up4_supported(cfg.dtype, ng, cfg.batch, bo.H, bo.W, fc.Cin, g0.C, fc.Cout_pad, fc.Kpad);
    if (up) {
      for (int g = 0; g < ng; ++g) {
        fuse[g].up_low = at(scr_set[1 + 2 * g].small);
        fuse[g].up_h = g0.mapH; fuse[g].up_w = g0.mapH; fuse[g].up_cs = g0.C;
      }
      for (int g = 0; g < ng; ++g) gvit_low.insert(gv[g].out);
    }
    gv_skip_up = up;
  }
  hipStream_t lane_g = sg == sm ? sm : fresh_side();
  TRY(order(sm, lane_g));
  stream = lane_g;
  int rcg = run_vit_g(ng, gv, 1);
  gv_skip_up = false;
  TRY(rcg);
  stream = sm;
  if (lv[0].v->shrink) {
    // v5:1139,1190: x = conv_shrink(x) ... x = conv_extend(x); both are per-pixel, so they run on the whole map around the windowed block
    ConvCall sh[3], ex[3];
    for (int g = 0; g < ng; ++g) {
      const std::string& ln = lv[g].out;
      sh[g] = ConvCall{ln + ".shrink", in[g], "", "", "", ln + ".shrunk", nullptr};
      ex[g] = ConvCall{ln + ".extend", ln + ".tok", "", "", "", ln, nullptr};
      lv[g].in = ln + ".shrunk";
      lv[g].out = ln + ".tok";
    }
    TRY(run_conv_g(ng, sh, 1));
    TRY(run_vit_g(ng, lv, 0));
    TRY(run_conv_g(ng, ex, 1));
  } else {
    TRY(run_vit_g(ng, lv, 0));
  }
  TRY(order(lane_g, sm));
  return run_conv_g(ng, fuse, 1);
}

// The three decoders (v3:546-697 / 706-853 / 862-1009) in lockstep: each layer of R, S and D is one grouped launch.
// D consumes R's and S's upsampled maps through CFSM2G (v3:885,920), which the lockstep order provides for free.
int cfen_net::run_decoder(float* const* outs, hipStream_t sm, hipStream_t sg) {
  const int dt = cfg.dtype, B = cfg.batch;
  static const char* tags = "rsd";
  bool tail_fused = false;
  stages_on_chip = false;
  tail_whole = false;
  auto tail_fusable = [&]() {
    if (cfs || !cfen_tune_tail_fused() || !an_pending.empty() || dt != CFEN_F16) return false;
    for (int g = 0; g < 3; ++g) {
      const std::string T = std::string("tail_") + (char)(tags[g] - 32);
      const ConvLayer& cu = convs.at(std::string("us_conv_d01") + tags[g]);
      const ConvLayer& cc = convs.at(T + ".conv3");
      const Buf& bi = bufs.at(std::string("lgcat_conv_d01") + tags[g]);
      if (!cu.tile || !cc.tile || !cfen_up_conv3_fused_supported(dt, bi.cs, cu.Cout_pad, bufs.at(std::string("us_conv_d01") + tags[g]).cs, cc.Cout_pad, bi.H, bi.W))
        return false;
    }
    return true;
  };
  auto names = [](const std::string& base, std::string* o) { for (int g = 0; g < 3; ++g) o[g] = base + tags[g]; };
  stream = sm;
  std::string in[3], out[3], up[3];
  // ---- level 3: block -> ConvT -> InstanceNorm -> ReLU (v3:301-302) ----
  for (int g = 0; g < 3; ++g) in[g] = "lgcat_conv_e03";
  names("lgcat_conv_d03", out);
  TRY(run_level_g(3, tags, 3, in, nullptr, out, sm, sg));
  names("us_conv_d03", up);
  {
    ConvCall c[3];
    for (int g = 0; g < 3; ++g) c[g] = ConvCall{up[g], out[g], "", "", "", up[g], nullptr};
    TRY(run_conv_g(3, c, 0));
    // the three maps are allocated back to back: one InstanceNorm over 3B "images"
    const Buf& bu = bufs.at(up[0]);
    const size_t bytes = (size_t)B * bu.H * bu.W * bu.cs * esz;
    CFEN_CHECK_ARG(bufs.at(up[1]).off == bu.off + bytes && bufs.at(up[2]).off == bu.off + 2 * bytes, "net: us_conv_d03 maps are not contiguous");
    label = "us_conv_d03 (x3):instnorm";
    TRYP(K_NORM, 0, cfen_instnorm_relu_impl(dt, map_ptr(up[0]), (float*)at(o_stats_set[0]), 3 * B, bu.H * bu.W, bu.C, bu.cs, 1e-5f, stream));
  }
  for (int l = 2; l >= 1; --l) {
    const std::string L = std::to_string(l), Lup = std::to_string(l + 1);
    // ---- skip fuse: D through CFSM2G over (D, R, S) upsampled maps, R and S through a 1x1 conv with the encoder skip ----
    const std::string cf = crs ? "sk_conv_d0" + Lup + "d" : "cfsm2g_d0" + Lup + "d";
    {
      const Buf& bu = bufs.at(cf);
      label = cf;
      if (crs) {   // crs:854,889: sk_conv_d0Xd(cat(D, R, S upsampled maps))
        ConvCall c3{cf, "us_conv_d0" + Lup + "d", "us_conv_d0" + Lup + "r", "", "", cf, nullptr, "us_conv_d0" + Lup + "s"};
        TRY(run_conv_g(1, &c3, 1));
      } else {
        TRYP(K_NORM, 0, cfen_cfsm2g_impl(dt, map_ptr("us_conv_d0" + Lup + "d"), map_ptr("us_conv_d0" + Lup + "r"), map_ptr("us_conv_d0" + Lup + "s"),
                                         map_ptr(cf), Pf(cf + ".w"), (float*)at(o_stats_set[1]), B, bu.H * bu.W, bu.C, bu.cs, stream));
      }
      ConvCall c[2];
      for (int g = 0; g < 2; ++g) {
        const std::string sk = "sk_conv_d0" + Lup + tags[g];
        c[g] = ConvCall{sk, "us_conv_d0" + Lup + tags[g], l == 2 ? "lgcat_conv_e02" : "lgcat_conv_e01", "", "", sk, nullptr};
        in[g] = sk;
      }
      TRY(run_conv_g(2, c, 1));
      in[2] = cf;
    }
    names("lgcat_conv_d0" + L, out);
    // `xr = us_conv_d01r(r_d_01 + xf)` (v3:696,852,1008): the extra `+ xf` rides along as a second residual of the
    // level-1 fuse conv, so stage lgcat_conv_d01* holds (reference stage + xf).
    TRY(run_level_g(3, tags, l, in, l == 1 ? (cfs ? "head" : "ds_conv_e01") : nullptr, out, sm, sg));
    if (cfs && l == 1) {   // cfs:669,823,977: the tails read d_01 + xf (folded into lgcat_conv_d01*'s epilogue) directly
      for (int g = 0; g < 3; ++g) up[g] = out[g];
      break;
    }
    names("us_conv_d0" + L, up);
    if (l == 1 && tail_fusable()) {
      tail_fused = true;      // us_conv_d01* runs inside the tails' first launch below (k_up_conv3_fused)
      break;
    }
    ConvCall c[3];
    for (int g = 0; g < 3; ++g) c[g] = ConvCall{up[g], out[g], "", "", "", up[g], nullptr};
    TRY(run_conv_g(3, c, 1));
  }
  // ---- tails: conv3 (+ActNorm for R, D) + ReLU, reflect-pad conv7 + tanh, fp32 NCHW out (v3:348-383) ----
  ConvCall c3[3], c7[3];
  for (int g = 0; g < 3; ++g) {
    const std::string T = std::string("tail_") + (char)(tags[g] - 32);
    c3[g] = ConvCall{T + ".conv3", up[g], "", "", "", T + ".mid", nullptr};
    c7[g] = ConvCall{T + ".conv7", T + ".mid", "", "", "", "", outs[g]};
  }
  if (tail_fused) {
    // ConvTranspose (24 -> 12, ActNorm, ReLU) + the tail's 3x3 in ONE grouped launch: the 12-channel full-resolution map between them stays in LDS
    CfenUpConv3 u[3];
    double fl = 0.0;
    for (int g = 0; g < 3; ++g) {
      const ConvLayer& cu = convs.at(up[g]);
      const ConvLayer& cc = convs.at(c3[g].layer);
      const Buf& bi = bufs.at(out[g]);
      u[g] = CfenUpConv3{map_ptr(out[g]), B, bi.H, bi.W, bi.cs, P(up[g] + ".wr"), Pf(up[g] + ".scale"), Pf(up[g] + ".shift"), 1,
                         P(c3[g].layer + ".wr"), Pf(c3[g].layer + ".scale"), Pf(c3[g].layer + ".shift"), 1, map_ptr(c3[g].out),
                         cfen_tune_keep_stages() ? map_ptr(up[g]) : nullptr};
      fl += B * (2.0 * cu.Cin_real * cu.Cout * 16.0 * bi.H * bi.W + 2.0 * cc.Cout * (double)cc.Cin_real * 9.0 * 4.0 * bi.H * bi.W);
    }
    // "net.tail_fused" = 2 (round 5): the reflect-pad 7x7 + tanh rides along too (k_tail.hip: a workgroup walks down a 64-column strip, both intermediate
    // maps in LDS) -- unless the stage maps are asked for
    bool whole = cfen_tune_tail_fused() == 2 && !cfen_tune_keep_stages();
    ConvDesc d7[3];
    for (int g = 0; g < 3 && whole; ++g) {
      const ConvLayer& z = convs.at(c7[g].layer);
      const Buf& bm = bufs.at(c7[g].in0);
      const int mode = output_u8 ? 2 : output_f16 ? 3 : 1;
      whole = z.tz && z.kind == 0 && z.k == 7 && z.reflect && outs[g] &&
              cfen_tail_fused_supported(dt, bufs.at(out[g]).cs, convs.at(up[g]).Cout_pad, bufs.at(up[g]).cs, convs.at(c3[g].layer).Cout_pad, bufs.at(out[g]).H,
                                        bufs.at(out[g]).W, z.Cout, mode) && bm.cs == 16;
      if (!whole) break;
      cfen_desc_conv(&d7[g], B, bm.H, bm.W, bm.cs, z.Cin, z.k, z.stride, z.pad, z.reflect, z.nsrc);
      d7[g].weight = P(c7[g].layer + ".wz"); d7[g].Kpad = z.Kpad;
      d7[g].scale = Pf(c7[g].layer + ".scale"); d7[g].shift = Pf(c7[g].layer + ".shift");
      d7[g].act = 2; d7[g].Cout = z.Cout; d7[g].Cout_pad = z.Cout_pad;
      d7[g].out = outs[g]; d7[g].out_nchw_f32 = mode; d7[g].cs_out = z.Cout_pad;
      fl += B * 2.0 * z.Cout * (double)z.Cin_real * 49.0 * bm.H * bm.W;
    }
    if (whole) {
      label = up[0] + " + " + c3[0].layer + " + " + c7[0].layer + " (x3, fused)";
      TRYP(K_CONV, fl, cfen_tail_fused_impl_g(dt, 3, u, d7, stream));
      stages_on_chip = true;
      tail_whole = true;
      return CFEN_OK;
    }
    label = up[0] + " + " + c3[0].layer + " (x3, fused)";
    TRYP(K_CONV, fl, cfen_up_conv3_fused_impl_g(dt, 3, u, stream));
    stages_on_chip = !cfen_tune_keep_stages();
  } else {
    TRY(run_conv_g(3, c3, 1));
  }
  return run_conv_g(3, c7, 2);
}

int cfen_net::forward(const float* x, float* xr, float* xs, float* xd) {
  const int dt = cfg.dtype, B = cfg.batch, N = cfg.load_size;
  const hipStream_t s0 = stream;
  const bool par = parallel && !profiling;
  if (par && !side[0]) {
    for (int k = 0; k < NSIDE; ++k)
      if (hipStreamCreateWithFlags(&side[k], hipStreamNonBlocking) != hipSuccess) {
        cfen_set_error("net: cannot create side stream");
        return CFEN_ERR_HIP;
      }
  }
  ev_next = 0;
  gvit_low.clear();
  side_next = 0;
  // two lanes: the caller's stream (CNN + LViT) and a side stream per level for GViT (drawn fresh per fork)
  const hipStream_t sg = par ? side[NSIDE - 1] : s0;   // only a marker "!= s0": run_level_g draws the real stream
  stream = s0;
  float* stats = (float*)at(o_stats_set[0]);
  // arrival counters of the split-K GEMMs: every launch leaves them zero, this makes a forward independent of whatever ran (or died) before -- where
  // the plan has such launches ("gemm.splitk", the persistent chains); the default plan has none and gets no zeroing launch (prime_workspace zeroed them once)
  const bool chains = gvit_stream && cfen_tune_gvit_chain() != 0;
  if (cfen_tune_gemm_splitk() || chains)
    for (int k = 1; k < 6; k += 2) TRY(cfen_zero_async(at(scr_set[k].splitk), CFEN_SPLITK_COUNTERS * sizeof(unsigned), stream));
  // ... and the grid-barrier words of the persistent GViT chains (the error word behind them is sticky: only cfen_net_chain_errors clears it)
  if (chains)
    for (int k = 1; k < 6; k += 2) TRY(cfen_zero_async(at(scr_set[k].sync), GV_ERR_WORD * sizeof(unsigned), stream));
  gv_launch = 0;
  launch_idx = 0;
  const Buf& bin = bufs.at("input");
  const bool use_head5 = head5 && cfen_tune_head5() && !cfen_tune_head_fused();
  if (use_head5) {
    // the input layout pass and head.0.0 in ONE launch: the conv stages its halo straight from the fp32 NCHW tensor / the uint8 HWC image
    const ConvLayer& c = convs.at("head.0.0");
    label = "head.0.0 (from the network input)";
    TRYP(K_CONV, B * 2.0 * c.Cout * (double)c.Cin_real * c.k * c.k * (double)full * full,
         cfen_head5_impl(dt, input_u8, x, P("head.0.0.w5"), Pf("head.0.0.scale"), Pf("head.0.0.shift"), map_ptr("head.conv5"), B, full, full,
                         bufs.at("head.conv5").cs, 0, stream));
  } else if (input_u8) {
    label = "input:u8hwc_to_nhwc";
    TRYP(K_TOKEN, 0, cfen_u8hwc_to_nhwc_impl(dt, (const unsigned char*)x, map_ptr("input"), B, full, full, bin.cs, stream));
  } else {
    label = "input:nchw_to_nhwc";
    TRYP(K_TOKEN, 0, cfen_nchw_to_nhwc_impl(dt, x, map_ptr("input"), B, 3, full, full, bin.cs, stream));
  }
  // head: conv5x5 + ResBlock                                                   (v3:123-127,395)
  const ConvLayer& h5 = convs.at("head.0.0");
  if (cfen_tune_head_fused() && h5.tile && convs.at("head.0.1.body.0").tile && convs.at("head.0.1.body.2").tile &&
      cfen_head_fused_supported(dt, bin.cs, h5.Cout, full, full)) {
    // one launch, the 5x5 output and the ResBlock's hidden map stay in LDS (k_conv_tile.hip: k_head_fused)
    label = "head (conv5 + ResBlock fused)";
    double fl = 0.0;
    for (const char* l : {"head.0.0", "head.0.1.body.0", "head.0.1.body.2"}) {
      const ConvLayer& c = convs.at(l);
      fl += B * 2.0 * c.Cout * (double)c.Cin_real * c.k * c.k * (double)full * full;
    }
    TRYP(K_CONV, fl, cfen_head_fused_impl(dt, map_ptr("input"), map_ptr("head"), P("head.0.0.wr"), Pf("head.0.0.scale"), Pf("head.0.0.shift"),
                                          P("head.0.1.body.0.wr"), Pf("head.0.1.body.0.scale"), Pf("head.0.1.body.0.shift"), P("head.0.1.body.2.wr"),
                                          Pf("head.0.1.body.2.scale"), Pf("head.0.1.body.2.shift"), B, full, full, stream));
  } else {
    if (!use_head5) TRY(run_conv("head.0.0", "input", nullptr, nullptr, nullptr, 0, "head.conv5", nullptr));
    const ConvLayer& ca = convs.at("head.0.1.body.0");
    const ConvLayer& cb = convs.at("head.0.1.body.2");
    if (cfen_tune_resblock_fused() && ca.tile && cb.tile && an_pending.empty() &&
        cfen_resblock_fused_supported(dt, bufs.at("head.conv5").cs, ca.Cout, full, full)) {
      // ResBlock in one launch: the hidden map (head.res_mid) stays in LDS (k_fuse.hip)
      label = "head.0.1 (ResBlock fused)";
      const double fl = B * 2.0 * (double)full * full * 9.0 * ((double)ca.Cout * ca.Cin_real + (double)cb.Cout * cb.Cin_real);
      TRYP(K_CONV, fl, cfen_resblock_fused_impl(dt, map_ptr("head.conv5"), map_ptr("head"), P("head.0.1.body.0.wr"), Pf("head.0.1.body.0.scale"),
                                               Pf("head.0.1.body.0.shift"), P("head.0.1.body.2.wr"), Pf("head.0.1.body.2.scale"),
                                               Pf("head.0.1.body.2.shift"), B, full, full, stream));
    } else {
      TRY(run_conv("head.0.1.body.0", "head.conv5", nullptr, nullptr, nullptr, 1, "head.res_mid", nullptr));
      TRY(run_conv("head.0.1.body.2", "head.res_mid", nullptr, "head.conv5", nullptr, 0, "head", nullptr));
    }
  }
  auto down = [&](const std::string& layer, const std::string& in) -> int {   // conv s2 -> IN -> ReLU (v3:292-298)
    TRY(run_conv(layer, in, nullptr, nullptr, nullptr, 0, layer, nullptr));
    const Buf& b = bufs.at(layer);
    label = layer + ":instnorm";
    TRYP(K_NORM, 0, cfen_instnorm_relu_impl(dt, map_ptr(layer), stats, B, b.H * b.W, b.C, b.cs, 1e-5f, stream));
    return CFEN_OK;
  };
  std::string in = "head";
  for (int l = 1; l <= 3; ++l) {
    const std::string L = std::to_string(l), out = "lgcat_conv_e0" + L;
    std::string ds = "ds_conv_e0" + L;
    if (cfs && l == 1)
      ds = "head";             // cfs:368: xf = head(input), level 1 runs at the image's own resolution
    else
      TRY(down(ds, in));
    TRY(run_level_g(1, "e", l, &ds, nullptr, &out, s0, sg));
    in = out;
  }
  float* outs[3] = {xr, xs, xd};
  TRY(run_decoder(outs, s0, sg));
  stream = s0;
  return CFEN_OK;
}

extern "C" {

int cfen_net_create(cfen_net** out, const cfen_net_config* cfg) {
  CFEN_CHECK_ARG(out && cfg, "net_create: null argument");
  cfen_net* n = new (std::nothrow) cfen_net();
  CFEN_CHECK_ARG(n != nullptr, "net_create: out of host memory");
  n->cfg = *cfg;
  int rc = n->build();
  if (rc) {
    delete n;
    return rc;
  }
  *out = n;
  return CFEN_OK;
}

void cfen_net_destroy(cfen_net* net) { delete net; }

size_t cfen_net_workspace_bytes(const cfen_net* net) { return net ? net->ws_bytes : 0; }

int cfen_net_set_param(cfen_net* net, const char* name, const void* dev_ptr, size_t nbytes) {
  CFEN_CHECK_ARG(net && name && dev_ptr, "set_param: null argument");
  auto it = net->params.find(name);
  CFEN_CHECK_ARG(it != net->params.end(), "set_param: unknown parameter '%s'", name);
  CFEN_CHECK_ARG(it->second.need == nbytes, "set_param: '%s' needs %zu bytes, got %zu", name, it->second.need, nbytes);
  CFEN_CHECK_ARG(cfen_aligned16(dev_ptr), "set_param: '%s' must be 16-byte aligned", name);
  it->second.ptr = dev_ptr;
  return CFEN_OK;
}

int cfen_net_actnorm_pending(cfen_net* net, const char* layer, const float* ones, const float* conv_bias, float* an_out) {
  CFEN_CHECK_ARG(net && layer && ones && conv_bias && an_out, "actnorm_pending: null argument");
  auto it = net->convs.find(layer);
  CFEN_CHECK_ARG(it != net->convs.end(), "actnorm_pending: unknown layer '%s'", layer);
  CFEN_CHECK_ARG(cfen_aligned16(ones) && cfen_aligned16(conv_bias) && cfen_aligned16(an_out), "actnorm_pending: pointers must be 16-byte aligned");
  const std::string ln(layer);
  const bool in_lvit = ln.size() > 7 && (ln.compare(ln.size() - 7, 7, ".shrink") == 0 || ln.compare(ln.size() - 7, 7, ".extend") == 0);
  net->an_pending[layer] = cfen_net::AnPending{ones, conv_bias, an_out, in_lvit ? net->cfg.patch_size : 0};
  return CFEN_OK;
}

int cfen_net_actnorm_pending_count(const cfen_net* net) { return net ? (int)net->an_pending.size() : 0; }

int cfen_net_set_input_u8(cfen_net* net, int enabled) {
  CFEN_CHECK_ARG(net != nullptr, "set_input_u8: null net");
  net->input_u8 = enabled ? 1 : 0;
  return CFEN_OK;
}

int cfen_net_set_output_u8(cfen_net* net, int enabled) {
  CFEN_CHECK_ARG(net != nullptr, "set_output_u8: null net");
  if (enabled)
    for (const char* t : {"tail_R.conv7", "tail_S.conv7", "tail_D.conv7"}) {
      auto it = net->convs.find(t);
      CFEN_CHECK_ARG(it != net->convs.end() && it->second.tz && (it->second.Cout == 1 || it->second.Cout == 3),
                     "set_output_u8: %s does not run on the Toeplitz 7x7 kernel at this geometry (fp16, image edge a multiple of 64): take fp32 outputs and cfen_tensor2im_u8", t);
    }
  CFEN_CHECK_ARG(!enabled || !net->output_f16, "set_output_u8: the net already writes fp16 outputs");
  net->output_u8 = enabled ? 1 : 0;
  return CFEN_OK;
}

int cfen_net_set_output_f16(cfen_net* net, int enabled) {
  CFEN_CHECK_ARG(net != nullptr, "set_output_f16: null net");
  if (enabled) {
    CFEN_CHECK_ARG(!net->output_u8, "set_output_f16: the net already writes uint8 outputs");
    for (const char* t : {"tail_R.conv7", "tail_S.conv7", "tail_D.conv7"}) {
      auto it = net->convs.find(t);
      CFEN_CHECK_ARG(it != net->convs.end() && it->second.tz, "set_output_f16: %s does not run on the fused tail kernel at this geometry (fp16, image edge a multiple of 64)", t);
    }
  }
  net->output_f16 = enabled ? 1 : 0;
  return CFEN_OK;
}

int cfen_net_missing_params(const cfen_net* net, char* buf, size_t buflen) {
  int count = 0;
  size_t pos = 0;
  if (buf && buflen) buf[0] = 0;
  if (!net) return 0;
  for (const auto& kv : net->params)
    if (!kv.second.ptr) {
      ++count;
      if (buf && pos + kv.first.size() + 2 < buflen) {
        memcpy(buf + pos, kv.first.c_str(), kv.first.size());
        pos += kv.first.size();
        buf[pos++] = ';';
        buf[pos] = 0;
      }
    }
  return count;
}

int cfen_net_forward(cfen_net* net, const float* x, float* xr, float* xs, float* xd, void* workspace, size_t workspace_bytes,
                     void* stream) {
  CFEN_CHECK_ARG(net && x && xr && xs && xd && workspace, "net_forward: null argument");
  if (workspace_bytes < net->ws_bytes) {
    cfen_set_error("net_forward: workspace has %zu bytes, %zu needed", workspace_bytes, net->ws_bytes);
    return CFEN_ERR_STATE;
  }
  CFEN_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "net_forward: workspace must be 256-byte aligned");
  for (const auto& kv : net->params)
    if (!kv.second.ptr) {
      cfen_set_error("net_forward: parameter '%s' was never set", kv.first.c_str());
      return CFEN_ERR_STATE;
    }
  net->base = (unsigned char*)workspace;
  net->stream = (hipStream_t)stream;
  TRY(net->prime_workspace(false));
  return net->forward(x, xr, xs, xd);
}

// Build the launch plan as an explicit hipGraph (kernel nodes + dependency edges; lanes that run on side
// streams in eager mode become parallel branches) and instantiate it.  Pointers are baked into the nodes.
int cfen_net_graph_capture(cfen_net* net, const float* x, float* xr, float* xs, float* xd, void* workspace, size_t workspace_bytes,
                           int32_t* graph_id) {
  CFEN_CHECK_ARG(net && x && xr && xs && xd && workspace && graph_id, "graph_capture: null argument");
  if (workspace_bytes < net->ws_bytes) {
    cfen_set_error("graph_capture: workspace has %zu bytes, %zu needed", workspace_bytes, net->ws_bytes);
    return CFEN_ERR_STATE;
  }
  CFEN_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "graph_capture: workspace must be 256-byte aligned");
  for (const auto& kv : net->params)
    if (!kv.second.ptr) {
      cfen_set_error("graph_capture: parameter '%s' was never set", kv.first.c_str());
      return CFEN_ERR_STATE;
    }
  CfenGraphRecorder rec;
  if (hipGraphCreate(&rec.graph, 0) != hipSuccess) {
    cfen_set_error("graph_capture: hipGraphCreate failed");
    return CFEN_ERR_HIP;
  }
  net->base = (unsigned char*)workspace;
  net->stream = (hipStream_t)(uintptr_t)16;          // lane key of the main lane; nothing is launched on it
  if (int prc = net->prime_workspace(true)) { (void)hipGraphDestroy(rec.graph); return prc; }
  cfen_recorder() = &rec;
  int rc = net->forward(x, xr, xs, xd);
  cfen_recorder() = nullptr;
  hipGraphExec_t exec = nullptr;
  if (rc == CFEN_OK && hipGraphInstantiate(&exec, rec.graph, nullptr, nullptr, 0) != hipSuccess) {
    cfen_set_error("graph_capture: hipGraphInstantiate failed (%zu nodes)", rec.nodes);
    rc = CFEN_ERR_HIP;
  }
  (void)hipGraphDestroy(rec.graph);
  if (rc) return rc;
  net->execs.push_back(exec);
  *graph_id = (int32_t)net->execs.size() - 1;
  return CFEN_OK;
}

int cfen_net_graph_launch(cfen_net* net, int32_t graph_id, void* stream) {
  CFEN_CHECK_ARG(net && graph_id >= 0 && (size_t)graph_id < net->execs.size(), "graph_launch: bad graph id");
  if (hipGraphLaunch(net->execs[graph_id], (hipStream_t)stream) != hipSuccess) {
    cfen_set_error("graph_launch: hipGraphLaunch failed");
    return CFEN_ERR_HIP;
  }
  return CFEN_OK;
}

int cfen_net_profile(cfen_net* net, const float* x, float* xr, float* xs, float* xd, void* workspace, size_t workspace_bytes, void* stream,
                     double* ms_per_class, double* flops_per_class, int32_t* launches_per_class, int nclass) {
  CFEN_CHECK_ARG(net && ms_per_class && flops_per_class && launches_per_class && nclass >= K_NCLASS, "net_profile: bad arguments");
  net->profiling = true;
  net->recs.clear();
  int rc = cfen_net_forward(net, x, xr, xs, xd, workspace, workspace_bytes, stream);
  net->profiling = false;
  for (int i = 0; i < nclass; ++i) { ms_per_class[i] = 0; flops_per_class[i] = 0; launches_per_class[i] = 0; }
  if (rc == CFEN_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
    cfen_set_error("net_profile: stream synchronize failed");
    rc = CFEN_ERR_HIP;
  }
  net->last_profile.clear();
  for (auto& r : net->recs) {
    float ms = 0.f;
    if (rc == CFEN_OK && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      ms_per_class[r.cls] += ms; flops_per_class[r.cls] += r.flops; launches_per_class[r.cls] += 1;
      r.ms = ms;
      net->last_profile.push_back(r);
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  net->recs.clear();
  return rc;
}

int cfen_net_profile_entry(const cfen_net* net, int index, const char** label, int32_t* kernel_class, double* flops, double* ms) {
  CFEN_CHECK_ARG(net && index >= 0, "net_profile_entry: bad arguments");
  if ((size_t)index >= net->last_profile.size()) return CFEN_ERR_STATE;   // past the end (not an error message: callers iterate until this)
  const cfen_net::Rec& r = net->last_profile[(size_t)index];
  if (label) *label = r.label.c_str();
  if (kernel_class) *kernel_class = r.cls;
  if (flops) *flops = r.flops;
  if (ms) *ms = r.ms;
  return CFEN_OK;
}

int cfen_net_profile_entry_kernel(const cfen_net* net, int index, const char** kernel, double* bytes) {
  CFEN_CHECK_ARG(net && index >= 0, "net_profile_entry_kernel: bad arguments");
  if ((size_t)index >= net->last_profile.size()) return CFEN_ERR_STATE;
  const cfen_net::Rec& r = net->last_profile[(size_t)index];
  if (kernel) *kernel = r.kernel.c_str();
  if (bytes) *bytes = r.bytes;
  return CFEN_OK;
}

int cfen_net_stage(const cfen_net* net, const char* name, const void** ptr, int32_t* C, int32_t* cs, int32_t* H, int32_t* W) {
  CFEN_CHECK_ARG(net && name && ptr, "net_stage: null argument");
  auto it = net->bufs.find(name);
  CFEN_CHECK_ARG(it != net->bufs.end(), "net_stage: unknown stage '%s'", name);
  if (!net->base) {
    cfen_set_error("net_stage: no forward has run yet");
    return CFEN_ERR_STATE;
  }
  if (net->gvit_low.count(name)) {
    cfen_set_error("net_stage: '%s' was never stored at full resolution in the last forward (its x4 upsampling ran inside the fuse conv); cfen_tune(\"net.keep_stages\", 1) stores it", name);
    return CFEN_ERR_STATE;
  }
  if (net->stages_on_chip && (!strncmp(name, "us_conv_d01", 11) || (net->tail_whole && !strncmp(name, "tail_", 5)))) {
    cfen_set_error("net_stage: '%s' stayed on chip in the last forward (fused into the tail's first launch); cfen_tune(\"net.keep_stages\", 1) stores it", name);
    return CFEN_ERR_STATE;
  }
  *ptr = net->base + it->second.off;
  if (C) *C = it->second.C;
  if (cs) *cs = it->second.cs;
  if (H) *H = it->second.H;
  if (W) *W = it->second.W;
  return CFEN_OK;
}

int cfen_net_chain_error_words(const cfen_net* net, const void** words, int n) {
  CFEN_CHECK_ARG(net && words && n >= 3, "net_chain_error_words: needs room for 3 pointers");
  if (!net->base) {
    cfen_set_error("net_chain_error_words: no forward has run yet");
    return CFEN_ERR_STATE;
  }
  for (int k = 0; k < 3; ++k) words[k] = (const unsigned*)net->at(net->scr_set[2 * k + 1].sync) + cfen_net::GV_ERR_WORD;
  return CFEN_OK;
}

double cfen_net_flops_per_image(const cfen_net* net) {
  if (!net) return 0.0;
  const int B = net->cfg.batch;
  double f = 0.0;
  for (const Vit& v : net->vits) {   // 2T(5D^2 + 4DH) + 4 T S D
    const double T = (double)(v.mapH / v.ws) * (v.mapH / v.ws) * v.S;
    f += 2.0 * T * (5.0 * v.Dn * v.Dn + 4.0 * (double)v.Dn * v.hidden) + 4.0 * T * v.S * v.Dn;
  }
  for (const auto& kv : net->convs) {
    const ConvLayer& c = kv.second;
    const double e = (double)c.out_edge;
    if (c.kind == 1)
      f += 2.0 * c.Cin_real * c.Cout * 16.0 * (e / 2.0) * (e / 2.0);   // 2*Cin*Cout*16*Hin*Win
    else
      f += 2.0 * c.Cout * (double)c.Cin_real * c.nsrc * c.k * c.k * e * e;
  }
  (void)B;
  return f;
}

}  // extern "C"
