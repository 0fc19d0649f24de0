// extern "C" surface of libcfen_hip.so (declared in include/cfen_hip.h): thin argument adapters over
// the kernel launchers.  No allocation, no synchronisation, no exceptions.
#include <cxxabi.h>
#include <ctype.h>
#include <unordered_map>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/cfen_hip.h"
#include "cfen_common.hpp"
#include "cfen_conv.hpp"
#include "cfen_internal.hpp"
#include "cfen_mlp.hpp"
#include "cfen_lvit.hpp"
#include <math.h>

static thread_local char g_err[512] = "";

void cfen_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

CfenGraphRecorder*& cfen_recorder() {
  static thread_local CfenGraphRecorder* rec = nullptr;
  return rec;
}
// zero `bytes` of device memory on a lane: eagerly, or as a memset node of the graph being built
// Zeroing of synchronisation words (split-K arrival counters, the chains' barrier words) in front of the launches that use them.
// MEASURED (round 4, MI355X, ROCm 7.2): as a hipGraph MEMSET node -- dependencies recorded like every kernel node's -- the zeroing is NOT ordered against its
// neighbouring kernel nodes once several instantiated graphs are in flight on different streams: counters were cleared in the middle of a launch that was
// counting arrivals (outputs off by 0.07 .. 0.2, no wait gave up), one graph at a time it is fine.  As a kernel node it is ordered.  "net.zero_memset" = 1 is the
// old form, kept for the A/B (tests/test_hip_net.py::test_three_forwards_in_flight...; profiles/r04_ab_zero_memset.txt).
int& cfen_tune_zero_memset() { static int v = 0; return v; }
int cfen_zero_async(void* p, size_t bytes, hipStream_t s) {
  CFEN_CHECK_ARG(p && bytes % 4 == 0, "zero_async: bad region");
  if (!cfen_tune_zero_memset()) return cfen_zero_words_impl(p, bytes / 4, s);   // a kernel (node): ordered like every other launch of the plan
  if (CfenGraphRecorder* rec = cfen_recorder()) {
    hipMemsetParams mp;
    memset(&mp, 0, sizeof(mp));
    mp.dst = p; mp.elementSize = 4; mp.width = bytes / 4; mp.height = 1; mp.pitch = bytes; mp.value = 0;
    std::vector<hipGraphNode_t>& deps = rec->tail[s];
    hipGraphNode_t node;
    if (hipGraphAddMemsetNode(&node, rec->graph, deps.empty() ? nullptr : deps.data(), deps.size(), &mp) != hipSuccess) {
      cfen_set_error("zero_async: hipGraphAddMemsetNode failed");
      return CFEN_ERR_HIP;
    }
    deps.assign(1, node);
    ++rec->nodes;
    return CFEN_OK;
  }
  if (hipMemsetAsync(p, 0, bytes, s) != hipSuccess) {
    cfen_set_error("zero_async: hipMemsetAsync failed");
    return CFEN_ERR_HIP;
  }
  return CFEN_OK;
}

std::string& cfen_kernel_log() {
  static thread_local std::string log;
  return log;
}

// "_ZN12_GLOBAL__N_110k_gemm_dmaIDF16_Li1ELi3ELi3EEEv7Grouped..." -> "k_gemm_dma<f16, 1, 3, 3>": the kernel's name and its template arguments where they are
// builtin types / integer / bool literals (every kernel of this library), "..." for anything else
static std::string cfen_demangle_lite(const char* sym) {
  std::string m(sym);
  size_t q = 0;
  if (m.rfind("_ZN12_GLOBAL__N_1", 0) == 0) q = 17;
  else if (m.rfind("_Z", 0) == 0) q = 2;
  else return m;
  size_t len = 0;
  while (q < m.size() && isdigit((unsigned char)m[q])) len = len * 10 + (m[q++] - '0');
  if (!len || q + len > m.size()) return m;
  std::string out = m.substr(q, len);
  q += len;
  if (q >= m.size() || m[q] != 'I') return out;
  ++q;
  out += "<";
  bool first = true;
  while (q < m.size() && m[q] != 'E') {
    std::string a;
    if (m.compare(q, 5, "DF16_") == 0) { a = "f16"; q += 5; }
    else if (m[q] == 'f') { a = "f32"; ++q; }
    else if (m[q] == 'd') { a = "f64"; ++q; }
    else if (m[q] == 'h') { a = "u8"; ++q; }
    else if (m[q] == 'i') { a = "int"; ++q; }
    else if (m[q] == 'L' && q + 2 < m.size()) {
      const char t = m[q + 1];
      size_t e = m.find('E', q);
      if (e == std::string::npos) { a = "..."; q = m.size(); }
      else {
        std::string v = m.substr(q + 2, e - q - 2);
        if (!v.empty() && v[0] == 'n') v[0] = '-';
        a = t == 'b' ? (v == "1" ? "true" : "false") : v;
        q = e + 1;
      }
    } else { out += first ? "..." : ", ..."; break; }
    out += (first ? "" : ", ") + a;
    first = false;
  }
  return out + ">";
}

// host stub -> "k_mlp3<24, 2, 3, 1536, 0, 1, 4, 0, 1, 0>" (see CFEN_LAUNCH)
void cfen_log_kernel(const void* host_stub, const char* as_written) {
  std::string& l = cfen_kernel_log();
  if (l.size() > 600) return;
  static thread_local std::unordered_map<const void*, std::string> names;
  auto it = names.find(host_stub);
  if (it == names.end()) {
    std::string n;
    const char* sym = hipKernelNameRefByPtr(host_stub, nullptr);
    if (sym && *sym) {
      int st = 0;
      char* dm = abi::__cxa_demangle(sym, nullptr, nullptr, &st);
      n = (st == 0 && dm) ? dm : cfen_demangle_lite(sym);       // (libstdc++'s demangler does not know DF16_ = _Float16)
      free(dm);
      int depth = 0;      // drop the parameter list: cut at the first '(' outside the template brackets that is not "(anonymous namespace)"
      for (size_t q = 0; q < n.size(); ++q) {
        if (n[q] == '<') ++depth;
        else if (n[q] == '>') --depth;
        else if (n[q] == '(' && depth == 0 && n.compare(q, 21, "(anonymous namespace)") != 0) { n.resize(q); break; }
      }
      if (n.rfind("void ", 0) == 0) n.erase(0, 5);
      for (const char* drop : {"(anonymous namespace)::", "_Float16", "__half"}) {
        const std::string d(drop), r = d[0] == '(' ? "" : "half_t";
        for (size_t q; (q = n.find(d)) != std::string::npos;) n.replace(q, d.size(), r);
      }
    }
    if (n.empty()) {
      n = as_written;
      while (!n.empty() && (n.front() == '(' || n.front() == ' ')) n.erase(n.begin());
      while (!n.empty() && (n.back() == ')' || n.back() == ' ')) n.pop_back();
    }
    it = names.emplace(host_stub, n).first;
  }
  const std::string& n = it->second;
  if (l.find(n) != std::string::npos) return;
  if (!l.empty()) l += " + ";
  l += n;
}

hipError_t& cfen_last_launch() {
  static thread_local hipError_t e = hipSuccess;
  return e;
}

extern "C" {

int cfen_abi_version(void) { return 1; }
const char* cfen_last_error(void) { return g_err; }

int cfen_gemm_nt(int dtype, const void* X, int ldx, const void* W, int ldw, const float* bias, const void* R, int ldr, const void* P,
                 int period, void* Y, int ldy, int M, int N, int K, int relu, void* stream) {
  return cfen_gemm_impl(dtype, X, ldx, W, ldw, bias, R, ldr, P, period, Y, ldy, M, N, K, relu, (hipStream_t)stream);
}

int cfen_gemm_ln(int dtype, const void* X, int ldx, const void* Wl, int ldw, const float* s, const float* bias, void* Y, int ldy, int M, int N,
                 int K, int relu, float eps, void* stream) {
  CFEN_CHECK_ARG(s != nullptr, "gemm_ln: the row sums of the folded weight are required");
  CFEN_CHECK_ARG(eps == cfen_gemm_lnf_eps(), "gemm_ln: eps must be %g (the generator's only LayerNorm eps)", (double)cfen_gemm_lnf_eps());
  const CfenGemmPtrs q{X, Wl, bias, nullptr, nullptr, Y, nullptr, s};
  return cfen_gemm_impl_g(dtype, 1, &q, ldx, ldw, 0, 1, ldy, M, N, K, relu, nullptr, (hipStream_t)stream, nullptr, 0);
}

int cfen_gemm_splitk(int dtype, const void* X, int ldx, const void* W, int ldw, const float* lnf_s, const float* bias, const void* R, int ldr, void* Y,
                      int ldy, int M, int N, int K, int relu, int nsplit, void* scratch, size_t scratch_bytes, void* stream) {
  CFEN_CHECK_ARG(nsplit >= 1 && scratch && cfen_aligned16(scratch), "gemm_splitk: needs nsplit >= 1 and an aligned scratch buffer");
  const CfenGemmPtrs q{X, W, bias, R, nullptr, Y, nullptr, lnf_s};
  float* ws = (float*)scratch;
  return cfen_gemm_impl_g(dtype, 1, &q, ldx, ldw, ldr, 1, ldy, M, N, K, relu, nullptr, (hipStream_t)stream, &ws, scratch_bytes, nullptr, nsplit);
}

int cfen_head_conv5(int dtype, int in_u8, const void* in, const void* w5, const float* scale, const float* shift, void* out, int B, int H, int W,
                    int cs_out, int act, void* stream) {
  return cfen_head5_impl(dtype, in_u8, in, w5, scale, shift, out, B, H, W, cs_out, act, (hipStream_t)stream);
}

int cfen_gemm_chain(int dtype, const cfen_chain_args* a, int team, void* stream) {
  CFEN_CHECK_ARG(a && a->sync_ws && a->sync_ws_bytes >= 8192 && cfen_aligned16(a->sync_ws), "gemm_chain: needs >= 8192 aligned bytes of synchronisation words");
  CFEN_CHECK_ARG(a->nphases >= 1 && a->nphases <= 5, "gemm_chain: 1..5 phases");
  CfenChainArgs c{};
  for (int p = 0; p < a->nphases; ++p) {
    const cfen_chain_phase& q = a->phase[p];
    c.ph[p] = CfenChainPhase{q.x, q.w_stream, q.bias, q.lnf_s, q.residual, q.pos, q.y, q.ldx, q.ldr, q.ldy, q.period, q.N, q.K, q.relu, q.nsplit, q.fold};
  }
  c.nph = a->nphases; c.M = a->M;
  c.fH = a->fold_H; c.fW = a->fold_W; c.fcs = a->fold_cs; c.fC = a->fold_C; c.fp = a->fold_p;
  unsigned* w = (unsigned*)a->sync_ws;
  c.bar = w; c.err = w + 1; c.cnt = w + 1024; c.ncnt = 1024;
  c.part = (float*)((unsigned char*)a->sync_ws + 8192); c.part_bytes = a->sync_ws_bytes - 8192;
  int rc = cfen_zero_async(a->sync_ws, 8192, (hipStream_t)stream);   // barrier word, error word, arrival counters
  if (rc) return rc;
  return cfen_gvit_chain_impl_g(dtype, 1, &c, team, (hipStream_t)stream);
}

int cfen_layernorm(int dtype, const void* X, void* Y, const float* gamma, const float* beta, int M, int D, float eps, void* stream) {
  CFEN_CHECK_ARG(gamma && beta, "layernorm: gamma/beta required");
  return cfen_layernorm_impl(dtype, X, Y, gamma, beta, M, D, eps, (hipStream_t)stream);
}

int cfen_attention(int dtype, const void* qkv, void* out, int nseq, int S, int heads, int dh, void* stream) {
  return cfen_attention_impl(dtype, qkv, out, nseq, S, heads, dh, (hipStream_t)stream);
}

int cfen_mlp_block(int dtype, const cfen_mlp_args* a, void* stream) {
  CFEN_CHECK_ARG(a != nullptr, "mlp_block: null args");
  MlpArgs m{};
  m.X = a->x; m.Y = a->y; m.fmap = a->fmap; m.ln_g = a->ln_gamma; m.ln_b = a->ln_beta;
  m.A = a->att; m.Wp = a->w_proj;
  m.W1a = a->w1a; m.b1a = a->b1a; m.W2a = a->w2a; m.b2a = a->b2a;
  m.W1b = a->w1b; m.b1b = a->b1b; m.W2b = a->w2b; m.b2b = a->b2b;
  m.M = a->M; m.D = a->D; m.H = a->H; m.eps = a->eps;
  m.mapH = a->mapH; m.mapW = a->mapW; m.C = a->C; m.cs = a->cs; m.ws = a->ws; m.p = a->p;
  return cfen_mlp_impl(dtype, &m, (hipStream_t)stream);
}

int cfen_mlp_stream_block(int dtype, const cfen_mlp_stream_args* a, void* stream) {
  CFEN_CHECK_ARG(a != nullptr, "mlp_stream_block: null args");
  Mlp3Args m{};
  m.X = a->x; m.A = a->att; m.Wp = a->wp_stream; m.Y = a->y; m.fmap = a->fmap; m.ln_g = a->ln_gamma; m.ln_b = a->ln_beta;
  m.Wa = a->wa_stream; m.b1a = a->b1a; m.b2a = a->b2a; m.Wb = a->wb_stream; m.b1b = a->b1b; m.b2b = a->b2b;
  m.M = a->M; m.D = a->D; m.H = a->H; m.eps = a->eps;
  m.mapH = a->mapH; m.mapW = a->mapW; m.C = a->C; m.cs = a->cs; m.ws = a->ws; m.p = a->p;
  return cfen_mlp3_impl_g(dtype, 1, &m, (hipStream_t)stream);
}

int cfen_lvit_window(int dtype, const cfen_lvit_args* a, void* stream) {
  CFEN_CHECK_ARG(a != nullptr, "lvit_window: null args");
  LvitArgs v{a->fmap, a->out, a->B, a->H, a->W, a->C, a->cs_in, a->cs_out, a->ws, a->p, a->w_stream, a->be, a->pos, a->ln1_gamma, a->ln1_beta,
             a->ln2_gamma, a->ln2_beta, a->b1a, a->b2a, a->b1b, a->b2b, a->hidden, a->eps, 1.4426950408889634f / sqrtf(24.f)};
  return cfen_lvit_window_impl_g(dtype, 1, &v, (hipStream_t)stream);
}

int cfen_patchify(int dtype, const void* fmap, void* tokens, int B, int H, int W, int C, int cs, int ws, int p, int pool, void* stream) {
  return cfen_patchify_impl(dtype, fmap, tokens, B, H, W, C, cs, ws, p, pool, 0, (hipStream_t)stream);
}

int cfen_unpatchify(int dtype, const void* tokens, void* fmap, int B, int H, int W, int C, int cs, int ws, int p, void* stream) {
  return cfen_patchify_impl(dtype, fmap, const_cast<void*>(tokens), B, H, W, C, cs, ws, p, 1, 1, (hipStream_t)stream);
}

int cfen_upsample4(int dtype, const void* small, void* out, int B, int h, int w, int C, int cs_in, int cs_out, void* stream) {
  return cfen_upsample4_impl(dtype, small, out, B, h, w, C, cs_in, cs_out, (hipStream_t)stream);
}

int cfen_nchw_to_nhwc(int dtype, const float* in, void* out, int B, int C, int H, int W, int cs, void* stream) {
  return cfen_nchw_to_nhwc_impl(dtype, in, out, B, C, H, W, cs, (hipStream_t)stream);
}

int cfen_embed_gather(int dtype, const void* fmap, int B, int H, int W, int C, int cs, int ws, int p, const void* weight, int ldw,
                      const float* bias, const void* pos, int period, void* Y, int ldy, void* stream) {
  CFEN_CHECK_ARG(B > 0 && ws > 0 && p > 0 && ws % p == 0 && H > 0 && W > 0 && H % ws == 0 && W % ws == 0, "embed_gather: bad geometry");
  CfenTokGather tg{fmap, B, H, W, C, cs, ws, p};
  const int tw = ws / p;
  return cfen_embed_gather_impl(dtype, &tg, weight, ldw, bias, pos, period, Y, ldy, B * (H / ws) * (W / ws) * tw * tw, (hipStream_t)stream);
}

int cfen_embed_qkv(int dtype, const cfen_embed_qkv_args* a, void* stream) {
  CFEN_CHECK_ARG(a != nullptr, "embed_qkv: null args");
  CFEN_CHECK_ARG(a->B > 0 && a->ws > 0 && a->p > 0 && a->ws % a->p == 0 && a->H > 0 && a->W > 0 && a->H % a->ws == 0 && a->W % a->ws == 0,
                 "embed_qkv: bad geometry");
  const int tw = a->ws / a->p;
  CfenEmbedQkvArgs q{a->fmap, a->B, a->H, a->W, a->C, a->cs, a->ws, a->p, a->we, a->be, a->pos, a->ln_gamma, a->ln_beta, a->wqkv, a->x1, a->qkv,
                     (long long)a->B * (a->H / a->ws) * (a->W / a->ws) * tw * tw, a->p * a->p * a->C, a->eps, a->head_major_heads};
  return cfen_embed_qkv_impl_g(dtype, 1, &q, (hipStream_t)stream);
}

int cfen_embed_qkv_stream(int dtype, const cfen_embed_qkv_args* a, void* stream) {
  CFEN_CHECK_ARG(a != nullptr, "embed_qkv_stream: null args");
  CFEN_CHECK_ARG(a->B > 0 && a->ws > 0 && a->p > 0 && a->ws % a->p == 0 && a->H > 0 && a->W > 0 && a->H % a->ws == 0 && a->W % a->ws == 0,
                 "embed_qkv_stream: bad geometry");
  const int tw = a->ws / a->p;
  CfenEmbedQkvArgs q{a->fmap, a->B, a->H, a->W, a->C, a->cs, a->ws, a->p, a->we, a->be, a->pos, a->ln_gamma, a->ln_beta, a->wqkv, a->x1, a->qkv,
                     (long long)a->B * (a->H / a->ws) * (a->W / a->ws) * tw * tw, a->p * a->p * a->C, a->eps, a->head_major_heads};
  return cfen_front3_impl_g(dtype, 1, &q, (hipStream_t)stream);
}

int cfen_attention_head_major(int dtype, const void* qkv, void* out, int nseq, int S, int heads, int dh, void* stream) {
  return cfen_attention_hm_impl_g(dtype, 1, &qkv, &out, nseq, S, heads, dh, (hipStream_t)stream);
}

int cfen_u8hwc_to_nhwc(int dtype, const unsigned char* in, void* out, int B, int H, int W, int cs, void* stream) {
  return cfen_u8hwc_to_nhwc_impl(dtype, in, out, B, H, W, cs, (hipStream_t)stream);
}
int cfen_tensor2im_u8(const float* in, unsigned char* out, int C, int H, int W, void* stream) {
  return cfen_tensor2im_u8_impl(in, out, C, H, W, (hipStream_t)stream);
}

int cfen_tune(const char* key, int value) {
  CFEN_CHECK_ARG(key != nullptr, "tune: null key");
  if (!strcmp(key, "gemm.kernel")) {
    CFEN_CHECK_ARG(value >= -1 && value <= 25, "tune: gemm.kernel must be -1 .. 25");
    cfen_tune_gemm_kernel() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "gemm.large") || !strcmp(key, "gemm.small") || !strcmp(key, "gemm.mid")) {
    CFEN_CHECK_ARG(value == 2 || value == 3 || value == 4 || value == 5 || value == 12 || value == 13 || value == 14 || value == 15 || value == 22 || value == 23 ||
                   value == 24 || value == 25 || value == 32 || value == 34 || value == 45 || value == 65,
                   "tune: %s must be tile 2 .. 5 (+10 / +20 for 3 / 4 LDS stages; 34 = tile 4 with 5 stages, 45 / 65 = tile 5 with 6 / 8 stages)", key);
    (key[5] == 'l' ? cfen_tune_gemm_large() : key[5] == 'm' ? cfen_tune_gemm_mid() : cfen_tune_gemm_small()) = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "gemm.big")) {
    CFEN_CHECK_ARG(value == 0 || value == 6, "tune: gemm.big is 0 (off) or 6 (192 x 128 tiles)");
    cfen_tune_gemm_big() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "convT.tpw")) {
    CFEN_CHECK_ARG(value >= 1, "tune: convT.tpw must be >= 1");
    cfen_tune_convT_tpw() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "conv7.tpw")) {
    CFEN_CHECK_ARG(value >= 1, "tune: conv7.tpw must be >= 1");
    cfen_tune_conv7_tpw() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "conv.wlds") || !strcmp(key, "conv.wlds_maxlog")) {
    CFEN_CHECK_ARG(value >= 0, "tune: %s must be >= 0", key);
    (key[9] ? cfen_tune_conv_wlds_maxlog() : cfen_tune_conv_wlds()) = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "gemm.big_min_tiles")) {
    CFEN_CHECK_ARG(value >= 1, "tune: gemm.big_min_tiles must be positive");
    cfen_tune_gemm_big_min_tiles() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "embed.lds")) {
    cfen_tune_embed_lds() = value & 7;
    return CFEN_OK;
  }
  if (!strcmp(key, "embed.stages")) {
    CFEN_CHECK_ARG(value >= 2 && value <= 5, "tune: embed.stages is 2 .. 5 ring stages of k_embed_qkv2 (D = 192)");
    cfen_tune_embed_stages() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "mlp3.tm192")) {
    CFEN_CHECK_ARG((value >= 2 && value <= 4) || value == 22 || value == 24 || value == 25 || value == 28, "tune: mlp3.tm192 is 2, 3 or 4 token tiles a wave (one workgroup a CU), 22 / 24 = 2 tiles at 256 registers on a 3- / 4-slot ring");
    cfen_tune_mlp3_tm192() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "embed.defer_refill")) {
    cfen_tune_embed_defer_refill() = value != 0;
    return CFEN_OK;
  }
  if (!strcmp(key, "gemm.defer_refill")) {
    cfen_tune_gemm_defer_refill() = value != 0;
    return CFEN_OK;
  }
  if (!strcmp(key, "lvit.debug")) {
    cfen_tune_lvit_debug() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "front3.debug")) {
    cfen_tune_front3_debug() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "mlp3.pair")) {
    cfen_tune_mlp3_pair() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "mlp3.debug")) {
    cfen_tune_mlp3_debug() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "gemm.m128")) {
    CFEN_CHECK_ARG(value == 0 || value == 2 || value == 32 || value == 3 || value == 4 || value == 14 || value == 34, "tune: gemm.m128 is 0 or a k_gemm_dma tile id");
    cfen_tune_gemm_m128() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "gemm.splitk_stages")) {
    CFEN_CHECK_ARG(value == 0 || value == 3, "tune: gemm.splitk_stages is 0 or 3");
    cfen_tune_gemm_splitk_stages() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "gemm.splitk_release")) { cfen_tune_gemm_splitk_release() = value; return CFEN_OK; }
  if (!strcmp(key, "gemm.splitk")) {
    cfen_tune_gemm_splitk() = value != 0;
    return CFEN_OK;
  }
  if (!strcmp(key, "net.ln_fold")) {
    cfen_tune_ln_fold() = value != 0;
    return CFEN_OK;
  }
  if (!strcmp(key, "net.fused_front_max_dim")) {
    cfen_tune_fused_front_max_dim() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "net.skip_classes")) {
    cfen_tune_skip_classes() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "mlp.small_tiles")) {
    cfen_tune_mlp_small_tiles() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "lvit.shape")) {
    cfen_tune_lvit_shape() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "net.lvit_window")) {
    cfen_tune_lvit_window() = value != 0;
    return CFEN_OK;
  }
  if (!strcmp(key, "net.fold_in_gemm")) {
    cfen_tune_fold_in_gemm() = value != 0;
    return CFEN_OK;
  }
  if (!strcmp(key, "net.attn_head_major")) {
    cfen_tune_attn_head_major() = value != 0;
    return CFEN_OK;
  }
  if (!strcmp(key, "dcn.tps")) {
    cfen_tune_dcn_tps() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "gemm.nt")) {
    cfen_tune_gemm_nt() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "dcn.tile")) {
    cfen_tune_dcn_tile() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "attn.hm_pair")) {
    cfen_tune_attn_hm_pair() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "net.head_fused")) {
    cfen_tune_head_fused() = value != 0;
    return CFEN_OK;
  }
  if (!strcmp(key, "net.stream_front")) {
    CFEN_CHECK_ARG(value >= 0 && value <= 2, "tune: net.stream_front is 0 (never), 1 (grouped decoder launches) or 2 (always)");
    cfen_tune_stream_front() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "net.stream_mlp192")) {
    cfen_tune_stream_mlp192() = value != 0;
    return CFEN_OK;
  }
  if (!strcmp(key, "net.stream_mlp")) {
    CFEN_CHECK_ARG(value >= 0 && value <= 2, "tune: net.stream_mlp is 0 (never), 1 (grouped decoder launches) or 2 (always)");
    cfen_tune_stream_mlp() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "net.gvit_dummy_wgs")) { cfen_tune_gvit_dummy_wgs() = value; return CFEN_OK; }
  if (!strcmp(key, "net.gvit_dummy_us")) { cfen_tune_gvit_dummy_us() = value; return CFEN_OK; }
  if (!strcmp(key, "net.skip_from")) { cfen_tune_skip_from() = value; return CFEN_OK; }
  if (!strcmp(key, "net.skip_to")) { cfen_tune_skip_to() = value; return CFEN_OK; }
  if (!strcmp(key, "net.extra_launches")) { cfen_tune_extra_launches() = value; return CFEN_OK; }
  if (!strcmp(key, "net.gvit_dummy_levels")) { cfen_tune_gvit_dummy_levels() = value; return CFEN_OK; }
  if (!strcmp(key, "net.gvit_dummy_stream")) { cfen_tune_gvit_dummy_stream() = value; return CFEN_OK; }
  if (!strcmp(key, "gvit.team")) {
    CFEN_CHECK_ARG(value >= 1 && value <= 85, "tune: gvit.team is 1 .. 85 workgroups per block (three blocks share the chip)");
    cfen_tune_gvit_team() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "net.gvit_stream")) {
    CFEN_CHECK_ARG(value >= 0 && value <= 2, "tune: net.gvit_stream is 0 (never), 1 (serial launch plan only) or 2 (every plan)");
    cfen_tune_gvit_stream() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "gvit.max_concurrent")) {
    CFEN_CHECK_ARG(value >= 1 && value <= 8, "tune: gvit.max_concurrent is 1 .. 8 forwards in flight");
    cfen_tune_gvit_max_concurrent() = value;
    return CFEN_OK;
  }
  if (!strcmp(key, "net.tail_fused")) { CFEN_CHECK_ARG(value >= 0 && value <= 2, "tune: net.tail_fused is 0, 1 (ConvTranspose + 3x3) or 2 (+ the 7x7)"); cfen_tune_tail_fused() = value; return CFEN_OK; }
  if (!strcmp(key, "tail.balance")) { cfen_tune_tail_balance() = value; return CFEN_OK; }
  if (!strcmp(key, "tail.debug")) { cfen_tune_tail_debug() = value; return CFEN_OK; }
  if (!strcmp(key, "tail.segments")) { CFEN_CHECK_ARG(value >= 1 && value <= 64, "tune: tail.segments is 1 .. 64"); cfen_tune_tail_segments() = value; return CFEN_OK; }
  if (!strcmp(key, "net.up_fused")) { cfen_tune_up_fused() = value != 0; return CFEN_OK; }
  if (!strcmp(key, "net.zero_memset")) { cfen_tune_zero_memset() = value != 0; return CFEN_OK; }
  if (!strcmp(key, "net.keep_stages")) { cfen_tune_keep_stages() = value != 0; return CFEN_OK; }
  if (!strcmp(key, "net.resblock_fused")) { cfen_tune_resblock_fused() = value != 0; return CFEN_OK; }
  if (!strcmp(key, "net.head5")) { cfen_tune_head5() = value != 0; return CFEN_OK; }
  if (!strcmp(key, "gvit.debug")) { cfen_tune_gvit_debug() = value; return CFEN_OK; }
  if (!strcmp(key, "net.gvit_chain")) { cfen_tune_gvit_chain() = value; return CFEN_OK; }   // 0 off, 1 every GViT block, 2 the grouped decoder launches only, 3 the encoder blocks only
  if (!strcmp(key, "net.embed_gather")) {
    cfen_tune_embed_gather() = value != 0;
    return CFEN_OK;
  }
  cfen_set_error("tune: unknown key '%s'", key);
  return CFEN_ERR_ARG;
}

int cfen_conv2d(int dtype, const cfen_conv_args* a, void* stream) {
  CFEN_CHECK_ARG(a != nullptr, "conv2d: null args");
  ConvDesc d;
  if (a->kind == 0) {
    CFEN_CHECK_ARG(a->k >= 1 && a->k <= 7 && a->stride >= 1 && a->stride <= 2 && a->nsrc >= 1 && a->nsrc <= 3 &&
                   a->k * a->k * a->nsrc <= CFEN_MAX_TAPS, "conv2d: unsupported kernel/stride/nsrc");
    CFEN_CHECK_ARG((a->nsrc < 2 || a->src1) && (a->nsrc < 3 || a->src2), "conv2d: src1 / src2 missing");
    cfen_desc_conv(&d, a->B, a->Hin, a->Win, a->cs_in, a->Cin, a->k, a->stride, a->pad, a->reflect, a->nsrc);
  } else if (a->kind == 1) {
    cfen_desc_convT4(&d, a->B, a->Hin, a->Win, a->cs_in, a->Cin);
  } else {
    cfen_set_error("conv2d: unknown kind %d", a->kind);
    return CFEN_ERR_ARG;
  }
  d.src[0] = a->src0; d.src[1] = a->src1; d.src[2] = a->src2;
  d.weight = a->weight; d.Kpad = a->Kpad;
  d.scale = a->scale; d.shift = a->shift; d.act = a->act;
  d.res[0] = a->res0; d.res[1] = a->res1; d.cs_res = a->cs_res;
  d.out = a->out; d.cs_out = a->cs_out; d.Cout_pad = a->Cout_pad; d.Cout = a->Cout;
  d.out_nchw_f32 = a->out_nchw_f32;
  if (a->wlayout == 2) {
    CFEN_CHECK_ARG(a->kind == 0 && a->k == 7 && a->nsrc == 1 && a->stride == 1 && a->pad == 3, "conv2d: Toeplitz layout is for 7x7 stride-1 same-size convolutions");
    return cfen_conv7_tz_impl_g(dtype, 1, &d, (hipStream_t)stream);
  }
  if (a->wlayout == 1 && a->kind == 1) return cfen_convT_tile_impl(dtype, &d, (hipStream_t)stream);
  if (a->wlayout == 1) {
    CFEN_CHECK_ARG(a->kind == 0 && a->nsrc == 1 && a->stride == 1 && a->pad == a->k / 2, "conv2d: rows layout needs a stride-1 same-size Conv2d");
    return cfen_conv_tile_impl(dtype, &d, a->k, (hipStream_t)stream);
  }
  CFEN_CHECK_ARG(a->wlayout == 0, "conv2d: unknown wlayout %d", a->wlayout);
  return cfen_conv_impl(dtype, &d, (hipStream_t)stream);
}

size_t cfen_stats_workspace(int B, int C) { return cfen_stats_workspace_bytes(B, C); }

int cfen_instnorm_relu(int dtype, void* x, float* stats_ws, int B, int HW, int C, int cs, float eps, void* stream) {
  return cfen_instnorm_relu_impl(dtype, x, stats_ws, B, HW, C, cs, eps, (hipStream_t)stream);
}

int cfen_cfsm2g(int dtype, const void* x0, const void* x1, const void* x2, void* out, const float* w, float* stats_ws, int B, int HW,
                int C, int cs, void* stream) {
  return cfen_cfsm2g_impl(dtype, x0, x1, x2, out, w, stats_ws, B, HW, C, cs, (hipStream_t)stream);
}

}  // extern "C"
