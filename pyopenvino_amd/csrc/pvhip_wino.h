// Internal interface of the Winograd F(2x2, 3x3) convolution (pvhip_wino.hip), used by pvhip_conv.hip.
#pragma once
#include <cstddef>

namespace pvhip {

// 3x3, stride 1, pad 1 ("same"), C a multiple of 4, and not switched off with PVHIP_CONV_WINOGRAD=0
bool   wino_eligible(int c, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int h, int w, int oh, int ow);
size_t wino_pack_elems(int k, int c);                                 // floats of the transformed weight panel (0: not applicable)
int    wino_pack(const float* w_oihw, float* u, int k, int c);       // enqueue the weight transform
int    wino_conv(const float* x, const float* u, float* y, int n, int c, int h, int w, int k_out, const float* bias, int act,
                 float act_lo, float act_hi, int out_channel_offset, int out_channels_total);

// F(4x4, 3x3): additionally H and W multiples of 4 and enough patches to fill the chip (PVHIP_CONV_WINOGRAD4=0 switches it off,
// =force drops the size rule)
bool   wino4_eligible(int c, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int h, int w, int oh, int ow, int n);
size_t wino4_pack_elems(int k, int c);
int    wino4_pack(const float* w_oihw, float* u, int k, int c);
int    wino4_conv(int m, const float* x, const float* u, float* y, int n, int c, int h, int w, int k_out, const float* bias, int act,
                  float act_lo, float act_hi, int out_channel_offset, int out_channels_total);

// F(2x2, 5x5) on the same kernel (m = 2): 5x5 / stride 1 / pad 2 layers with even extents; its panel has wino4_pack_elems floats
bool   wino25_eligible(int c, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int h, int w, int oh, int ow, int n);
int    wino25_pack(const float* w_oihw, float* u, int k, int c);

// Pointwise (1x1 / stride 1 / unpadded, C a multiple of 16) convolutions (pvhip_pw.hip): weights in MFMA-fragment order, one
// workgroup per pixel tile and group of <= 8 32-channel tiles; PVHIP_CONV_POINTWISE=0 selects the general kernel
struct PwDest {
    float* y;
    int    m_begin, k;      // first panel row of this destination's channels, real channels in it
    int    ctotal, coff;    // channels of the tensor y points into, first channel of this convolution in it
};
bool   pw_eligible(int c, int kh, int kw, int sh, int sw, int pad_top, int pad_left, int h, int w, int oh, int ow);
size_t pw_pack_elems(int k, int c);
int    pw_pack(const float* w_oihw, float* ap, int k, int c);
int    pw_conv(const float* x, const float* ap, int n, int c, int hw, int k_panel, const float* bias, int act, float act_lo, float act_hi,
               int ndest, const PwDest* dests);

}  // namespace pvhip
