// Descriptor of one implicit-GEMM convolution launch (see k_conv.hip) and builders for the layer
// kinds of the v3 generator.  Host-only plain data; passed to the kernel by value.
#pragma once
#include <stdint.h>
#include <string.h>

#define CFEN_MAX_TAPS 64

struct ConvTap {
  int8_t dy, dx;   // input offset relative to base*in_stride
  int8_t src;      // 0..2: which input tensor (a 1x1 conv over a channel concat reads two or three)
  int8_t pad_;
};

struct ConvDesc {
  const void* src[3];
  int Hin, Win, cs_in, Cin;      // Cin = channels consumed per tap (<= cs_in)
  const void* weight;            // [nphase][Cout_pad][Kpad], k = tap*Cin + c, zero padded
  int Kpad;
  int ntaps, nphase;
  ConvTap taps[CFEN_MAX_TAPS];   // phase p uses taps[p*ntaps .. p*ntaps+ntaps)
  int in_stride, out_stride;
  int8_t ph_y[4], ph_x[4];       // output offset of each phase
  int B, Hb, Wb;                 // base grid (one GEMM column per base pixel per phase)
  int Hout, Wout;
  int pad_reflect;
  const float* scale;            // [Cout_pad] epilogue y = acc*scale + shift
  const float* shift;
  int act;                       // 0 none, 1 ReLU, 2 tanh
  const void* res[2];            // optional residual maps (same geometry as out, channel stride cs_res)
  int cs_res;
  void* out;
  int cs_out, Cout_pad, Cout;
  int out_nchw_f32;
  // optional (k_conv, 1x1 over a concat, LDS-staged weights): source 1 is GViT's LOW-resolution map [B][up_h][up_w][up_cs], up_h = Hin / 4; the kernel
  // applies upsam(upsam(.)) (v3:1323, k_upsample4's arithmetic) to the pixels of its workgroup in LDS instead of reading a full-resolution copy
  int up4, up_h, up_w, up_cs;
};

static inline int cfen_round_up(int v, int m) { return (v + m - 1) / m * m; }

// Conv2d(k, stride, padding=pad) over `nsrc` same-shaped inputs concatenated on channels.
static inline void cfen_desc_conv(ConvDesc* d, int B, int Hin, int Win, int cs_in, int Cin, int k, int stride, int pad, int reflect,
                                  int nsrc) {
  memset(d, 0, sizeof(*d));
  d->B = B; d->Hin = Hin; d->Win = Win; d->cs_in = cs_in; d->Cin = Cin;
  d->nphase = 1; d->in_stride = stride; d->out_stride = 1;
  d->Hout = (Hin + 2 * pad - k) / stride + 1;
  d->Wout = (Win + 2 * pad - k) / stride + 1;
  d->Hb = d->Hout; d->Wb = d->Wout;
  d->pad_reflect = reflect;
  int n = 0;
  for (int s = 0; s < nsrc; ++s)
    for (int ky = 0; ky < k; ++ky)
      for (int kx = 0; kx < k; ++kx) {
        d->taps[n].dy = (int8_t)(ky - pad); d->taps[n].dx = (int8_t)(kx - pad); d->taps[n].src = (int8_t)s;
        ++n;
      }
  d->ntaps = n;
}

// ConvTranspose2d(k=4, stride=2, padding=1): out[2b+p] gathers 2 taps per axis.
//   p=0: ky=1 (iy=b), ky=3 (iy=b-1);  p=1: ky=0 (iy=b+1), ky=2 (iy=b)        [oy = 2*iy - 1 + ky]
// Phase index = py*2+px; tap order inside a phase = (ty, tx) with the (ky, kx) listed above; the host
// packer (packing.py: pack_convT) emits weights in the same order.
static inline void cfen_desc_convT4(ConvDesc* d, int B, int Hin, int Win, int cs_in, int Cin) {
  memset(d, 0, sizeof(*d));
  d->B = B; d->Hin = Hin; d->Win = Win; d->cs_in = cs_in; d->Cin = Cin;
  d->nphase = 4; d->ntaps = 4; d->in_stride = 1; d->out_stride = 2;
  d->Hout = 2 * Hin; d->Wout = 2 * Win; d->Hb = Hin; d->Wb = Win;
  static const int8_t off[2][2] = {{0, -1}, {1, 0}};   // [parity][tap] -> input offset
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      int ph = py * 2 + px;
      d->ph_y[ph] = (int8_t)py; d->ph_x[ph] = (int8_t)px;
      for (int ty = 0; ty < 2; ++ty)
        for (int tx = 0; tx < 2; ++tx) {
          ConvTap* t = &d->taps[ph * 4 + ty * 2 + tx];
          t->dy = off[py][ty]; t->dx = off[px][tx]; t->src = 0;
        }
    }
}

#if defined(__HIPCC__)
// The scalar fields of a ConvDesc, copied out of the kernel-argument block once (one bulk scalar load): through the reference every use
// re-read its field (39 s_loads in the 1x1 fuse conv), and `src[tap.src]` -- a per-lane index -- was a VECTOR load of the pointer from the
// argument block in front of every pixel fetch, i.e. a second memory round trip per K chunk.
struct ConvK {
  const void *src0, *src1, *src2, *weight, *res0, *res1;
  void* out;
  const float *scale, *shift;
  int Hin, Win, cs_in, Cin, Kpad, ntaps, in_stride, out_stride, B, Hb, Wb, Hout, Wout, pad_reflect, act, cs_res, cs_out, Cout_pad, Cout, out_nchw_f32;
  int oy_off, ox_off;
  int up_h, up_w, up_cs;
};
__device__ __forceinline__ ConvK conv_k(const ConvDesc& r, int phase) {
  ConvK k;
  k.src0 = r.src[0]; k.src1 = r.src[1]; k.src2 = r.src[2]; k.weight = r.weight; k.res0 = r.res[0]; k.res1 = r.res[1]; k.out = r.out;
  k.scale = r.scale; k.shift = r.shift;
  k.Hin = r.Hin; k.Win = r.Win; k.cs_in = r.cs_in; k.Cin = r.Cin; k.Kpad = r.Kpad; k.ntaps = r.ntaps; k.in_stride = r.in_stride;
  k.out_stride = r.out_stride; k.B = r.B; k.Hb = r.Hb; k.Wb = r.Wb; k.Hout = r.Hout; k.Wout = r.Wout; k.pad_reflect = r.pad_reflect;
  k.act = r.act; k.cs_res = r.cs_res; k.cs_out = r.cs_out; k.Cout_pad = r.Cout_pad; k.Cout = r.Cout; k.out_nchw_f32 = r.out_nchw_f32;
  k.oy_off = r.ph_y[phase]; k.ox_off = r.ph_x[phase];
  k.up_h = r.up_h; k.up_w = r.up_w; k.up_cs = r.up_cs;
  return k;
}
#endif
