// Internal launcher prototypes shared by cfen_api.cpp and cfen_net.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct ConvDesc;

int cfen_gemm_impl(int dtype, const void* X, int ldx, const void* W, int ldw, const float* bias, const void* R, int ldr, const void* P,
                   int period, void* Y, int ldy, int M, int N, int K, int relu, hipStream_t s);
int cfen_zero_words_impl(void* p, size_t nwords, hipStream_t s);   // k_tokens.hip
int& cfen_tune_zero_memset();
int cfen_zero_async(void* p, size_t bytes, hipStream_t s);   // memset 0 on a lane (eager, or a node of the graph being recorded)
float& cfen_gemm_lnf_eps();   // LayerNorm eps used by the LN-folded GEMMs (1e-5, the only value the generator uses)
// Y = tok W^T + bias + tok + P[m % period]  with tok = the patch tokens of an NHWC map, gathered by the GEMM's loader
// (window partition + unfold + linear_encoding + residual + position add in one launch; v3:1025-1056,1140-1143,1166)
struct CfenTokGather { const void* map; int B, H, W, C, cs, ws, p; };
int cfen_embed_gather_impl(int dtype, const CfenTokGather* tg, const void* W, int ldw, const float* bias, const void* P, int period,
                           void* Y, int ldy, int M, hipStream_t s);
// LViT front half fused (k_embed.hip): gather + linear_encoding + residual + position + LayerNorm + qkv projection
struct CfenEmbedQkvArgs {
  const void* fmap; int B, H, W, C, cs, ws, p;        // NHWC map the patch tokens are gathered from
  const void* We; const float* be; const void* pos;   // [D][D] (k axis in packing.kperm32 order for fp16), [D], [S][D]
  const float* ln_g; const float* ln_b;
  const void* Wqkv;                                   // [3D][D], k axis as We
  void* X1; void* QKV;                                // [M][D], [M][3D]
  long long M; int D; float eps;
  int hm_heads;   // > 0: QKV is written head-major, [(window * heads + head) * 3 + {q,k,v}][S][D / heads] (k_attention_hm's input)
};
bool cfen_embed_qkv_supported(int D);
int cfen_embed_qkv_impl_g(int dtype, int ng, const CfenEmbedQkvArgs* a, hipStream_t s);
// the same front half for D = 384 on the fragment-stream ring (k_stream.hip); We / Wqkv are packing.pack_stream_rows streams
bool cfen_front3_supported(int dtype, int D, long long M);
int cfen_front3_impl_g(int dtype, int ng, const CfenEmbedQkvArgs* a, hipStream_t s);
// grouped launches (cfen_common.hpp: CFEN_MAX_GROUPS problems of identical geometry, one launch)
struct CfenGemmPtrs {
  const void* X; const void* W; const float* bias; const void* R; const void* P; void* Y; const void* gmap;
  // LayerNorm folded into the GEMM (k_gemm_dma only): X is the un-normalised row x, W = W0 * gamma (columns scaled on the host),
  // lnf_s[n] = sum_k W[n][k], bias = W0 beta + b0:  Y = act(rstd_m (x W^T - mean_m lnf_s) + bias) + ...; mean / rstd of every row are
  // accumulated by the workgroup from the X tiles it stages anyway.  null = plain GEMM.
  const float* lnf_s;
  // W is stored tile-major instead of row-major: [ceil(N / 96)][K * sizeof(T) / 128][96 rows][128 bytes], rows past N zero (packing.pack_wtile).
  // One K-step of a 96-feature tile is then ONE contiguous 12 KB run of HBM instead of 96 pieces of 128 bytes a whole row apart --
  // what the few-token GViT GEMMs (weights streamed once from HBM, 0.5 GB per forward) are bound by.  k_gemm_dma with 96-feature tiles only.
  int wtile = 0;
  void* ymap = nullptr;   // fold: Y goes into this NHWC map (geometry = cfen_gemm_impl_g's `yg`), see GemmArgs::ymap
};
int cfen_gemm_impl_g(int dtype, int ng, const CfenGemmPtrs* gp, int ldx, int ldw, int ldr, int period, int ldy, int M, int N, int K, int relu,
                     const CfenTokGather* tg, hipStream_t s, float* const* splitk_ws, size_t splitk_ws_bytes, const CfenTokGather* yg = nullptr,
                     int force_nsplit = 0);
// tg: geometry only, the maps are gp[g].gmap.  splitk_ws (may be null): one scratch per problem for split-K: CFEN_SPLITK_COUNTERS arrival
// counters (unsigned, ZERO before the first use; every launch leaves them zero) followed by the fp32 partial slabs.
// force_nsplit: 0 = shape rule, 1 = never split, > 1 = exactly this many K slices (tests)
constexpr int CFEN_SPLITK_COUNTERS = 1024;
int cfen_patchify_impl_g(int dtype, int ng, const void* const* fmap, void* const* tok, int B, int H, int W, int C, int cs, int ws, int p, int pool,
                         int inverse, hipStream_t s);
int cfen_upsample4_impl_g(int dtype, int ng, const void* const* small, void* const* out, int B, int h, int w, int C, int cs_in, int cs_out,
                          hipStream_t s);
int cfen_attention_impl_g(int dtype, int ng, const void* const* qkv, void* const* out, int nseq, int S, int heads, int dh, hipStream_t s);
int cfen_layernorm_impl_g(int dtype, int ng, const void* const* X, void* const* Y, const float* const* g, const float* const* b, int M, int D,
                          float eps, hipStream_t s, int Dn = 0);   // Dn: real entries per row when the rows carry zero padding slots (0 = D)
// window attention on the head-major qkv layout (see CfenEmbedQkvArgs::hm_heads); fp16, head_dim 24, S in {64, 256}
bool cfen_attention_hm_supported(int dtype, int S, int dh);
int cfen_attention_hm_impl_g(int dtype, int ng, const void* const* qkv, void* const* out, int nseq, int S, int heads, int dh, hipStream_t s);
int cfen_attention_impl(int dtype, const void* qkv, void* out, int nseq, int S, int heads, int dh, hipStream_t s);
int cfen_layernorm_impl(int dtype, const void* X, void* Y, const float* g, const float* b, int M, int D, float eps, hipStream_t s);
int cfen_patchify_impl(int dtype, const void* fmap, void* tok, int B, int H, int W, int C, int cs, int ws, int p, int pool, int inverse,
                       hipStream_t s);
int cfen_upsample4_impl(int dtype, const void* small, void* out, int B, int h, int w, int C, int cs_in, int cs_out, hipStream_t s);
int cfen_nchw_to_nhwc_impl(int dtype, const float* in, void* out, int B, int C, int H, int W, int cs, hipStream_t s);
int cfen_tensor2im_u8_impl(const float* in, unsigned char* out, int C, int H, int W, hipStream_t s);
int cfen_u8hwc_to_nhwc_impl(int dtype, const unsigned char* in, void* out, int B, int H, int W, int cs, hipStream_t s);
int cfen_conv_impl(int dtype, const ConvDesc* d, hipStream_t s);
// LDS-tiled stride-1 path (k_conv_tile.hip); weights in the "rows" layout, d->Kpad == cfen_conv_tile_kpad
bool cfen_conv_tile_supported(int dtype, int kind, int k, int stride, int pad, int nsrc, int cs_in, int Cout_pad, int H, int W);
int cfen_conv_tile_kpad(int dtype, int k, int cs_in);
int cfen_conv_tile_impl(int dtype, const ConvDesc* d, int k, hipStream_t s);
// head conv5x5 + ResBlock in one launch, intermediates in LDS (k_conv_tile.hip: k_head_fused); fp16, 12 head channels
bool cfen_head_fused_supported(int dtype, int cs_in, int C, int H, int W);
int cfen_head_fused_impl(int dtype, const void* in, void* out, const void* w5, const float* s5, const float* t5, const void* wa, const float* sa,
                         const float* ta, const void* wb, const float* sb, const float* tb, int B, int H, int W, hipStream_t s);
int& cfen_tune_head_fused();        // 1: the head runs as one k_head_fused launch where it applies; 0 (default, faster: see cfen_net.cpp) ("net.head_fused")
bool cfen_convT_tile_supported(int dtype, int cs_in, int Cout_pad, int Hin, int Win);
int cfen_convT_tile_kpad(int dtype, int cs_in);
int cfen_convT_tile_impl(int dtype, const ConvDesc* d, hipStream_t s);
int cfen_conv_impl_g(int dtype, int ng, const ConvDesc* d, hipStream_t s);
bool cfen_conv_up4_supported(int dtype, int ng, int B, int H, int W, int Cin, int cs_low, int Cout_pad, int Kpad);
int cfen_conv_tile_impl_g(int dtype, int ng, const ConvDesc* d, int k, hipStream_t s);
int cfen_convT_tile_impl_g(int dtype, int ng, const ConvDesc* d, hipStream_t s);
// 7x7, <= 4 output channels, fp32 NCHW output: Toeplitz-expanded weights [16][7][10 taps][16] (k_conv_tile.hip: k_conv7_tz)
bool cfen_conv7_tz_supported(int dtype, int k, int stride, int pad, int nsrc, int cs_in, int Cout, int out_nchw_f32, int H, int W);
int cfen_conv7_tz_kpad();
int cfen_conv7_tz_impl_g(int dtype, int ng, const ConvDesc* d, hipStream_t s);
size_t cfen_stats_workspace_bytes(int B, int C);
int cfen_instnorm_relu_impl(int dtype, void* x, float* part, int B, int HW, int C, int cs, float eps, hipStream_t s);
// ActNorm2d first-call init from the raw layer output x (NHWC, B images): per-channel statistics over the batch -> folded
// (scale, shift) epilogue table + raw (weight, bias) in an_out[2][Cpad]                      (models/actnorm.py:25-37)
int cfen_actnorm_init_impl(int dtype, const void* x, float* part, int B, int HW, int C, int cs, int Cpad, const float* conv_bias, float* scale,
                           float* shift, float* an_out, hipStream_t s);
int cfen_cfsm2g_impl(int dtype, const void* x0, const void* x1, const void* x2, void* out, const float* w, float* part, int B, int HW,
                     int C, int cs, hipStream_t s);
int& cfen_tune_gemm_kernel();   // -1 auto, 0 tiled, 1 skinny (cfen_tune "gemm.kernel")
int& cfen_tune_convT_tpw();     // "convT.tpw"
int& cfen_tune_conv7_tpw();     // tiles per workgroup of the Toeplitz 7x7 kernel ("conv7.tpw")
int& cfen_tune_conv_wlds();     // gather convs stage their weight matrix in LDS: 0 never, 1 matrices up to 60 KB, 2 up to 150 KB ("conv.wlds"; "conv.wlds_maxlog" caps the launch size)
int& cfen_tune_conv_wlds_maxlog();
int& cfen_tune_gemm_big();      // 192 x 128 tile for many-token GEMMs with N >= 768: 0 off (default: it is slower), 6 on ("gemm.big")
int& cfen_tune_gemm_big_min_tiles();   // ... when the launch has at least this many such tiles ("gemm.big_min_tiles")
int& cfen_tune_gemm_large();    // k_gemm_dma tile id (2..5) for problems with >= 1024 tiles of 96 x 64 ("gemm.large")
int& cfen_tune_gemm_small();
int& cfen_tune_gemm_mid();    // ... and for smaller ones ("gemm.small")
int& cfen_tune_embed_gather();  // 1 (default): LViT embedding gathers its tokens from the map; 0: separate k_patchify ("net.embed_gather")
int& cfen_tune_mlp_small_tiles();   // fused-MLP tiling ("mlp.small_tiles"): 0 256/128 tokens per 4-wave WG at 1 wave/SIMD, 1 half-size token tiles at
                                    // 2 waves/SIMD, 2 as 1 but TM = 2 for D = 192 (register-capped), 3 (default) 8-wave WGs: half the weight re-streaming
int& cfen_tune_skip_classes();    // bit mask of kernel classes NOT launched by the net (marginal-cost timing; outputs invalid) ("net.skip_classes")
int& cfen_tune_ln_fold();               // 1: LN1 / LN2 of the blocks without a fused kernel ride on the qkv / ffn1 GEMM ("net.ln_fold")
int& cfen_tune_fused_front_max_dim();   // k_embed_qkv is used for LViT embedding dims <= this (0 = never) ("net.fused_front_max_dim")
int& cfen_tune_gemm_splitk_stages();   // ring depth of the split-K tile beyond two stages: 0 or 3 ("gemm.splitk_stages")
int& cfen_tune_mlp3_tm192();         // token tiles per wave of k_mlp3 at D = 192: 2, 3 (default) or 4 ("mlp3.tm192")
int& cfen_tune_embed_defer_refill(); // k_embed_qkv2 likewise ("embed.defer_refill")
int& cfen_tune_gemm_defer_refill();  // k_gemm_dma: refill behind the K-step's first fragment reads ("gemm.defer_refill")
int& cfen_tune_lvit_debug();         // k_lvit_window ("lvit.debug"): 64 = section stamps of workgroup 0 to stderr
int& cfen_tune_mlp3_pair();          // D = 384 MLP blocks on the wave-pair kernel k_mlp3p ("mlp3.pair")
int& cfen_tune_front3_debug();       // k_front3 timing experiments ("front3.debug"): 1 no refills, 8 no qkv stores, 24 no stores (results invalid); | 64 = section stamps to stderr
int& cfen_tune_mlp3_debug();         // k_mlp3 timing experiments (results invalid): 1 no DMA refills, 2 no MFMAs ("mlp3.debug")
int& cfen_tune_gemm_m128();          // tile id (+10 per extra stage) for problems of <= 128 tokens, 0 = shape rule ("gemm.m128")
int& cfen_tune_gemm_nt();         // weight rows of k_gemm_dma by non-temporal LDS-DMA: 0 / 1 (M <= 512) / 2 ("gemm.nt")
int& cfen_tune_gemm_splitk();     // 1 (default): K-heavy few-token GEMMs run split-K when the caller provides scratch ("gemm.splitk")
int& cfen_tune_lvit_window();        // 1 (default): LViT level 1 runs as one k_lvit_window launch per instance group ("net.lvit_window")
int& cfen_tune_fold_in_gemm();      // 1 (default): the last GEMM of an unfused block folds its tokens into the map itself ("net.fold_in_gemm")
int& cfen_tune_attn_head_major();   // 1 (default): LViT levels with a fused front half hand qkv to attention head-major ("net.attn_head_major")
int& cfen_tune_dcn_tps();           // taps per K slice of k_dcn_lean at most this ("dcn.tps")
int& cfen_tune_dcn_tile();          // 1: deformable conv forward on k_dcn_tile where its shapes allow ("dcn.tile")
int& cfen_tune_attn_hm_pair();      // 1: S = 256 head-major attention on the two-query-tile kernel ("attn.hm_pair")
int& cfen_tune_stream_front();      // k_front3 for the D = 384 LViT blocks: 0 never, 1 (default) grouped decoder launches, 2 always ("net.stream_front")
int& cfen_tune_stream_mlp192();     // 1: LViT level 2 (D = 192) on k_mlp3 instead of k_mlp2; 0 (default, faster inside the forward: see cfen_net.cpp) ("net.stream_mlp192")
int& cfen_tune_stream_mlp();        // k_mlp3 for the D = 384 LViT blocks: 0 never, 1 (default) grouped decoder launches, 2 always ("net.stream_mlp")
int& cfen_tune_embed_lds();        // k_embed_qkv weights through LDS: bit 0 for D = 96, bit 1 for D = 192 ("embed.lds")
// what-if probe: a launch that holds `wgs` x `ng` CUs for `usec` microseconds ("net.gvit_dummy_*": CU-time experiments, results invalid)
int cfen_occupy_impl(int wgs, int ng, int usec, int do_stream, const void* src, size_t src_bytes, void* sink, hipStream_t s);
int& cfen_tune_gvit_dummy_wgs();
int& cfen_tune_gvit_dummy_us();
int& cfen_tune_gvit_dummy_stream();
int& cfen_tune_gvit_dummy_levels();
int& cfen_tune_extra_launches();
int& cfen_tune_skip_from();
int& cfen_tune_skip_to();
// Persistent GEMM chain (k_gvit.hip): up to 5 dependent GEMM phases Y = epi(X W^T) run by ONE launch of `team` workgroups per problem that
// meet at a grid barrier between phases.  W: fragment streams (packing.pack_stream_tiles).  fp16.
struct CfenChainPhase {
  const void* X; const void* W; const float* bias; const float* lnf_s; const void* R; const void* P; void* Y;
  int ldx, ldr, ldy, period, N, K, relu, nsplit, fold;
};
struct CfenChainArgs {
  CfenChainPhase ph[5];
  int nph, M;
  int fH, fW, fcs, fC, fp;     // fold geometry of the phases with fold = 1 (Y = NHWC map of fH x fW pixels, fp x fp patches of fC channels at stride fcs)
  unsigned* bar;               // grid-barrier counter, ZERO before the launch (one word per launch)
  unsigned* cnt; int ncnt;     // split-K arrival counters (zero; every launch leaves them zero)
  float* part; size_t part_bytes;   // split-K slabs
  unsigned* err;               // set to 1 when a wait gave up (never expected)
};
size_t cfen_gvit_chain_part_bytes(int M, int maxN, int max_nsplit);
int cfen_gvit_chain_impl_g(int dtype, int ng, const CfenChainArgs* ca, int team, hipStream_t s);
int& cfen_tune_gvit_team();     // workgroups per GViT block of the persistent chain ("gvit.team")
int& cfen_tune_gvit_max_concurrent();   // forwards of the chain plan that may be in flight at once ("gvit.max_concurrent", default 1): the teams of ALL of them must fit the chip
int& cfen_tune_gvit_chain();    // 1 (default): GViT blocks run their GEMMs as persistent chains where the net holds fragment-stream weights ("net.gvit_chain")
int& cfen_tune_gvit_debug();    // timing experiments on the chain kernel, results invalid ("gvit.debug")
int& cfen_tune_gemm_splitk_release();   // A/B: release fence in every split-K slice ("gemm.splitk_release")
// head.0.0 (conv 5x5, 3 -> 12 channels) read straight from the fp32 NCHW input or the uint8 HWC image (k_head5.hip); w5 = packing.pack_head5
bool cfen_head5_supported(int dtype, int Cout_pad, int cs_out, int H, int W);
int cfen_head5_impl(int dtype, int in_u8, const void* in, const void* w5, const float* scale, const float* shift, void* out, int B, int H, int W,
                    int cs_out, int act, hipStream_t s);
int& cfen_tune_head5();   // 1 (default): the input layout pass and head.0.0 run as one k_head5 launch where it applies ("net.head5")
// head.0.1 ResBlock (conv3x3 + ReLU + conv3x3 + skip) in one launch, the hidden map in LDS (k_fuse.hip); fp16, 16-channel-stride maps
bool cfen_resblock_fused_supported(int dtype, int cs, int C, int H, int W);
int cfen_resblock_fused_impl(int dtype, const void* in, void* out, const void* wa, const float* sa, const float* ta, const void* wb, const float* sb,
                             const float* tb, int B, int H, int W, hipStream_t s);
int& cfen_tune_resblock_fused();   // 1 (default): the head's ResBlock runs as one k_resblock_fused launch where it applies ("net.resblock_fused")
// us_conv_d01* (ConvTranspose 24 -> 12 + ActNorm + ReLU) + the tail's 3x3 in one grouped launch, the map between them in LDS (k_fuse.hip)
struct CfenUpConv3 {
  const void* in; int B, Hin, Win, cs_in;                      // (B, Hin, Win, cs_in) fp16 map, cs_in * 2 <= 64 bytes
  const void* wT; const float* sT; const float* tT; int actT;  // ConvTranspose: "<layer>.wr" [4][16][4 x 32], epilogue table, activation
  const void* w3; const float* s3; const float* t3; int act3;  // 3x3: "<layer>.wr" [16][3][2][32]
  void* out;                                                   // (B, 2 Hin, 2 Win, 16) output of the 3x3
  void* up_out;                                                // optional: the ConvTranspose output map (same shape), written for parity tests
};
bool cfen_up_conv3_fused_supported(int dtype, int cs_in, int Cup_pad, int cs_up, int C3_pad, int Hin, int Win);
int cfen_up_conv3_fused_impl_g(int dtype, int ng, const CfenUpConv3* u, hipStream_t s);
// the whole tail in one launch: ConvTranspose + 3x3 + reflect-pad 7x7 + tanh, both intermediate maps in LDS (k_tail.hip); d7 = the 7x7's descriptor as for cfen_conv7_tz_impl_g
struct ConvDesc;
bool cfen_tail_fused_supported(int dtype, int cs_in, int Cup_pad, int cs_up, int C3_pad, int Hin, int Win, int Cout7, int out_mode);
int cfen_tail_fused_impl_g(int dtype, int ng, const CfenUpConv3* u, const ConvDesc* d7, hipStream_t s);
int& cfen_tune_tail_balance();   // work split between k_tail_fused's wave groups ("tail.balance")
int& cfen_tune_tail_debug();     // timing experiments of k_tail_fused ("tail.debug", results invalid)
int& cfen_tune_tail_segments();  // vertical segments per 64-column strip of k_tail_fused ("tail.segments", default 1)
int& cfen_tune_up_fused();       // 1: GViT's x4 bilinear runs inside the level's fuse conv (k_conv UP), no k_upsample4 launch ("net.up_fused").  Default 0: measured
                                 // 6 launches and 0.35 GB of HBM traffic fewer per forward but 0.7 % SLOWER (the 9-tap interpolation per pixel on the vector
                                 // pipe in front of a K = 48 .. 192 1x1 costs more than the copy it saves: lgcat_conv_d01 72 -> 110 us for a 19 us launch)
int& cfen_tune_tail_fused();     // "net.tail_fused": 2 (default) the whole tail as one k_tail_fused launch, 1 us_conv_d01* + tail conv3 as one k_up_conv3_fused launch, 0 three launches
int& cfen_tune_keep_stages();    // 1: fused launches also store the stage maps they keep on chip (us_conv_d01*), for parity tests ("net.keep_stages"; default 0)
// out (B, h, w, cs_out) = 4 x 4 mean of in (B, 4h, 4w, cs_in): GViT's avgpool . avgpool as a map (k_tokens.hip: k_pool4)
int cfen_pool4_impl_g(int dtype, int ng, const void* const* in, void* const* out, int B, int h, int w, int C, int cs_in, int cs_out, hipStream_t s);
int& cfen_tune_embed_stages();   // LDS-DMA ring stages of k_embed_qkv2 at D = 192 ("embed.stages": 2 .. 5, default 4)
int& cfen_tune_gvit_stream();   // 1 (default): GViT blocks of embedding dim 384 run on the LViT-3 stream kernels (k_front3 / k_mlp3): 5 launches a block ("net.gvit_stream")
