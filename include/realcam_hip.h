/*
 * realcam_hip.h -- C ABI of librealcam_hip.so: the MI355X (gfx950) RAW->sRGB hot path.
 *
 * Drop-in boundary.  The upstream project (kepengxu/RealCamNet) has no FFI
 * layer: its hot path is a chain of stock ATen calls made from nn.Module.forward().  Each entry
 * point below replaces one such call site; the citation gives the reference file:line (paths
 * relative to the upstream repo).  The Python modules in realcamnet_amd/ keep the reference class
 * names / state_dict keys / forward() signatures and reach these symbols through ctypes.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer named d_* / in desc structs is DEVICE memory on
 *     the current HIP device unless the comment says "host".
 *   - activations are NHWC ("pixel-major"): element (b,y,x,c) at ((b*H + y)*W + x)*C + c.
 *   - dtype: RC_F32 (float) or RC_BF16 (bfloat16 storage, fp32 accumulate).
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*); no host sync, no
 *     allocation -> HIP-graph capturable.
 *   - return value: 0 on success, negative rc_status otherwise; rc_last_error() gives the text.
 *     Shape / alignment violations are reported, never silently "fixed".
 */
#ifndef REALCAM_HIP_H
#define REALCAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RC_ABI_VERSION 14

typedef enum rc_status {
    RC_OK = 0,
    RC_ERR_INVALID = -1,     /* bad argument (shape, alignment, null pointer, unsupported combination) */
    RC_ERR_HIP = -2,         /* a HIP runtime call failed */
    RC_ERR_UNSUPPORTED = -3  /* valid request, but no kernel instantiation covers it */
} rc_status;

typedef enum rc_dtype { RC_F32 = 0, RC_BF16 = 1, RC_U16 = 2 /* sensor counts: rc_raw_ingest's mosaic only */ } rc_dtype;

typedef enum rc_act { RC_ACT_NONE = 0, RC_ACT_RELU = 1, RC_ACT_LEAKY = 2 /* slope in act_slope */,
                      RC_ACT_GELU = 3 /* exact erf GELU: nn.GELU() in groupmix.Mlp */,
                      RC_ACT_RELU_POST = 4 /* ReLU applied LAST, after the residual add: relu(conv(x) + residual), the
                                              ResidualUnit of CompressAI's AttentionBlock (models/tcm.py:270 SWAtten) */ } rc_act;

typedef enum rc_out_mode {
    RC_OUT_NHWC = 0,           /* out[b][y][x][cout]                                             */
    RC_OUT_PIXEL_SHUFFLE2 = 1, /* nn.PixelShuffle(2) folded into the store:
                                  out[b][2y+i][2x+j][c] <- conv channel 4c+2i+j; out is NHWC (2H,2W,cout/4) */
    RC_OUT_NCHW = 2,           /* planar out[b][cout][y][x], cropped to (out_h,out_w); the network's
                                  final tensor (reference forward returns NCHW)                   */
    RC_OUT_PIXEL_SHUFFLE2_NCHW = 3, /* nn.PixelShuffle(2) + planar store: out[b][c][2y+i][2x+j] <- conv channel 4c+2i+j, cropped to
                                  (out_h,out_w) <= (2 height, 2 width); out_dtype fp32 or bf16.  The folded tail's store
                                  (rc_tail_fold_weights)                                          */
    RC_OUT_NHWC_DWT = 4        /* (ABI 14) the convolution followed by networks.DWTForward (models/networks.py:224-235, the `conv -> DWT`
                                  end of LiteISP's down1, models/LiteISP.py:1950-1953) in ONE launch: out is NHWC (height/2, width/2, 4 cout),
                                  out[b][y][x][4c+k] = sum_ij haar[k][i][j] * bf16(conv[b][2y+i][2x+j][c]) with the reference's frozen taps
                                  haar = .5 * {++++, ++--, +-+-, +--+}; the full-resolution map is never written.  Bit-identical to rc_conv2d
                                  (RC_OUT_NHWC) + rc_dwt_forward with those taps.  bf16, ksize 3, one Cin chunk and one cout tile of 32 or
                                  48 channels (the layers of the wave-autonomous kernel), even height / width, act NONE / RELU / LEAKY, or act NONE with a
                                  residual (NHWC, the conv's own shape: an RCAGroup's closing conv + group skip in front of the DWT, LiteISP down2);
                                  no film / mul_plus1 / gate / chan_sums: anything else is RC_ERR_UNSUPPORTED */
} rc_out_mode;

/* ---- library -------------------------------------------------------------------------------- */
int rc_abi_version(void);
const char* rc_last_error(void);           /* thread-local, host string */
const char* rc_build_info(void);           /* "gfx950 <compiler>" */
/* Device query used by the host mirror to fail loudly on a wrong GPU: writes the gcnArchName. */
int rc_device_arch(char* buf, size_t buflen);

/* ---- a1/a2: Bayer pixel-unshuffle + zero pad -------------------------------------------------
 * Replaces: the "Unpixel shuffle" box of assets/networkarch.png (README.md:33-38; upstream has no
 * code for it, every model already takes the packed tensor, models/LiteISP.py:2670) fused with
 * pad_to_multiple_of_16 (models/LiteISP.py:84-105).
 * mosaic: (B, 2*h, 2*w) one plane, dtype `in_dtype`; packed: NHWC (B, hp, wp, 4), hp>=h, wp>=w,
 * channel k = 2*i + j <- mosaic pixel (2y+i, 2x+j); rows/cols beyond (h,w) are written as zero. */
int rc_bayer_unshuffle(const void* d_mosaic, int in_dtype, void* d_packed, int out_dtype,
                       int batch, int h, int w, int hp, int wp, void* stream);

/* ---- f4: RAW ingest in front of the path ----------------------------------
 * The "Unpixel shuffle" and "Resize" boxes of assets/networkarch.png with the sensor normalisation upstream leaves to
 * the data loader: v' = (v - black_level) / (white_level - black_level); packed as rc_bayer_unshuffle (zero padded to
 * hp x wp); cond (B,4,cond_h,cond_w) NCHW = bilinear resize of the un-padded normalised packed RAW (h x w) with
 * torch.nn.functional.interpolate(mode="bilinear", align_corners=False) semantics -- the colour prior's input
 * (models/LiteISP.py:2016, x[1]).  One launch.  in_dtype may also be RC_U16 (sensor counts, e.g. black 64 / white 1023). */
int rc_raw_ingest(const void* d_mosaic, int in_dtype, void* d_packed, void* d_cond, int out_dtype, int batch, int h, int w,
                  int hp, int wp, int cond_h, int cond_w, float black_level, float white_level, void* stream);

/* ---- layout plumbing at the nn.Module boundary (reference tensors are NCHW) ------------------
 * nchw (B,C,h,w) -> nhwc (B,hp,wp,C) with zero padding (hp>=h, wp>=w) and dtype conversion. */
int rc_nchw_to_nhwc(const void* d_src, int src_dtype, void* d_dst, int dst_dtype,
                    int batch, int c, int h, int w, int hp, int wp, void* stream);
/* nhwc (B,H,W,C) -> nchw (B,C,h,w) cropping to h<=H, w<=W. */
int rc_nhwc_to_nchw(const void* d_src, int src_dtype, void* d_dst, int dst_dtype,
                    int batch, int c, int H, int W, int h, int w, void* stream);

/* ---- a3/a4/a7/a9/a11: KxK convolution, stride 1, "same" zero padding, MFMA implicit GEMM ------
 * Replaces: nn.Conv2d as built by networks.conv mode 'C' (models/networks.py:146-160) and its
 * fused neighbours: ReLU / LeakyReLU (networks.py:189-200), the RCAB/RCAGroup/U-Net additive skips
 * (networks.py:311,335; models/LiteISP.py:2028-2031), Res_GFM's FiLM + LeakyReLU(0.01)
 * (models/LiteISP.py:553-558), head*(lsc+1) (models/LiteISP.py:2014), nn.PixelShuffle(2)
 * (models/LiteISP.py:1998) and the CALayer input/gate (networks.py:267-270).
 *
 * Weights are re-packed once (host side) into the MFMA fragment order with rc_conv_pack_weights.
 */
typedef struct rc_conv_desc {
    int32_t batch, height, width;   /* input spatial size == conv output size                     */
    int32_t cin, cout, ksize;       /* ksize: 1 or 3 (zero padding ksize/2); 5 (cout <= 16; bf16 with cin % 48 == 0 or cin % 32 == 0, fp32 with cin % 16 == 0:
                                       the folded tail, rc_tail_fold_weights);
                                       2 (bf16, cin % 16 == 0, RC_OUT_NHWC only): the 2x2 window
                                       at pixel offsets {-1, 0}^2, weights (cout, cin, 2, 2) -- the non-zero taps of a stride-2 3x3
                                       convolution (compressai conv3x3(stride=2), ResidualBlockWithStride; models/tcm.py:336-345) taken
                                       over the rc_space_to_depth2 map of its input (9 of the 16 (tap, phase) weight blocks non-zero;
                                       the 3x3 embedding of the same convolution carries 36 blocks) */
    int32_t dtype;                  /* rc_dtype of activations + packed weights                   */
    /* input x.  in_gate==NULL: x = in0.
     * in_gate!=NULL (CALayer gate + RCAB skip, networks.py:270,311): x = in0*gate[b][c] + in1 and,
     * if in_store!=NULL, x is also written there (it is the next block's skip tensor).           */
    const void* in0;
    const void* in1;
    const float* in_gate;           /* (B, cin) fp32                                              */
    void* in_store;
    const void* wpacked;            /* device copy of rc_conv_pack_weights output                 */
    const float* bias;              /* device, packed order (rc_conv_pack_bias), or NULL          */
    /* epilogue on v = acc + bias, in this order:
     *   film_scale!=NULL: v = v*scale[b][c] + shift[b][c] + v        (Res_GFM, LiteISP.py:556)
     *   act                                                          (none / relu / leaky)
     *   mul_plus1!=NULL : v = v * (mul_plus1[b][y][x][c] + 1)        (LiteISP.py:2014)
     *   out_scale!=NULL : v = v * out_scale[b][c]                    (networks.py:270, last field)
     *   residual!=NULL  : v = v + residual[b][y][x][c]                                         */
    const float* film_scale;        /* (B, cout) fp32                                             */
    const float* film_shift;        /* (B, cout) fp32                                             */
    int32_t act;                    /* rc_act                                                     */
    float act_slope;
    const void* mul_plus1;          /* NHWC (B,H,W,cout), activation dtype                        */
    const void* residual;           /* NHWC (B,H,W,cout), activation dtype                        */
    void* out;
    int32_t out_mode;               /* rc_out_mode                                                */
    int32_t out_dtype;              /* rc_dtype of `out` (RC_OUT_NCHW may emit fp32 from a bf16 net;
                                       other modes require out_dtype == dtype)                    */
    int32_t out_h, out_w;           /* RC_OUT_NCHW crop size (<= height,width), RC_OUT_PIXEL_SHUFFLE2_NCHW crop size (<= 2 height, 2 width); else ignored */
    /* optional per-channel partial sums of v (the value stored), for CALayer's global mean
     * (networks.py:268): fp32 (B, rc_conv_sum_tiles(), cout), reduced in fixed order by rc_ca_gate */
    float* chan_sums;
    /* ksize 2 only, 0 = unused: in0 is the stride-2 convolution's OWN input (B, src_h, src_w, cin / 4) and the kernel gathers the
     * space-to-depth channels while staging (channel p*(cin/4) + k of map pixel (y, x) = channel k of source pixel (2y + (p >> 1),
     * 2x + (p & 1)), zero beyond the source edge -- rc_space_to_depth2's order), so no space-to-depth pass is launched.  Needs
     * height = ceil(src_h / 2), width = ceil(src_w / 2), cin / 4 a multiple of 64, no gated input.
     * With src_h set the layer IS a stride-2 3x3 convolution: the 7 of 16 (tap, phase) weight blocks such a convolution never touches (tap row -1 with
     * phase row 0, tap column -1 with phase column 0) are taken as zero and their multiplications are not issued, whatever the weight tensor holds there. */
    int32_t src_h, src_w;
    /* optional (B, cout) fp32: v = v * out_scale[b][c], applied after act / mul_plus1 and BEFORE the residual add -- the CALayer gate of this
     * conv's own output when it is known ahead of the launch (rc_ca_gate_ahead): RCABlock's x + CA(conv(...)) (networks.py:311, 270) leaves
     * the second conv as ONE map, x_new = conv2(t) * gate + x, instead of r = conv2(t) followed by r * gate + x in the next layer's staging. */
    const float* out_scale;
    /* ABI 10: partial-sum slots per image the caller allocated for chan_sums: rc_conv_sum_slots(desc) (what this launch fills -- the carried-sums
     * kernels write one slot per (residue class of their tile walk, wave): 2 048 per image at 4K instead of 32 640) or 0 / rc_conv_sum_tiles() = the
     * per-tile layout every kernel can write.  Any other value is an error. */
    int32_t chan_sums_slots;
    /* ABI 11 (the slot was reserved0): cout tile width in channels, 0 = automatic (64 / 48 / 80 / 16 by shape).  A multiple of 16 <= 80: narrower tiles give the
     * general kernel more blocks on maps too small to fill the chip (fp32 128 -> 128 at 135 x 240, B = 1: 272 blocks of 64 couts for 256 CUs; 16-wide tiles: 1 088).
     * The packed order depends on it: pack weights and bias with rc_conv_pack_weights_ct / rc_conv_pack_bias_ct and the SAME value.  Plain NHWC / NCHW stores, ksize 1 / 3. */
    int32_t cout_tile;
    /* ABI 12: 0 = the implicit GEMM (every shape above).  1 = Winograd F(2x2, 3x3) (csrc/wino.hip): the same stride-1 3x3 convolution with 2.25x fewer
     * multiplications -- 16 products M_xi = U_xi V_xi per 2x2 output tile, U = G g G^T packed by rc_wino_pack_weights (NOT rc_conv_pack_weights), V = B^T d B and
     * Y = A^T M A formed in registers.  fp32, ksize 3, cin % 8 == 0, cout % 16 == 0, RC_OUT_NHWC; bias / film vectors in NATURAL channel order (no rc_conv_pack_bias);
     * epilogue: bias, film, act (none / relu / leaky / relu_post), out_scale, residual, chan_sums (rc_conv_sum_slots(desc) slots); no gated input, mul_plus1, GELU,
     * cout_tile or src_h.  Anything else returns RC_ERR_UNSUPPORTED.  Results differ from algo 0 by fp32 rounding only (exact on small-integer data). */
    int32_t algo;
} rc_conv_desc;

/* Size in bytes of the packed weight buffer for (cin,cout,ksize,dtype,out_mode); 0 on error. */
size_t rc_conv_packed_bytes(int cin, int cout, int ksize, int dtype, int out_mode);
/* Host-side repack.  w_oihw: host fp32 (cout,cin,k,k) as in the state_dict; dst: host buffer of
 * rc_conv_packed_bytes() bytes (then copied to the device by the caller). */
int rc_conv_pack_weights(const float* w_oihw_host, int cin, int cout, int ksize, int dtype,
                         int out_mode, void* dst_host);
/* Packed-order length of bias / film vectors (cout rounded up to the kernel's cout tile) and the
 * host-side permutation: dst[j] = bias[perm(j)] or 0 for padding lanes. */
int rc_conv_packed_cout(int cin, int cout, int ksize, int dtype, int out_mode);
int rc_conv_pack_bias(const float* bias_host, int cin, int cout, int ksize, int dtype, int out_mode,
                      float* dst_host);
/* The same three with a caller-chosen cout tile width (rc_conv_desc.cout_tile; 0 = the functions above). */
size_t rc_conv_packed_bytes_ct(int cin, int cout, int ksize, int dtype, int out_mode, int cout_tile);
int rc_conv_packed_cout_ct(int cin, int cout, int ksize, int dtype, int out_mode, int cout_tile);
int rc_conv_pack_weights_ct(const float* w_oihw_host, int cin, int cout, int ksize, int dtype, int out_mode, int cout_tile, void* dst_host);
int rc_conv_pack_bias_ct(const float* bias_host, int cin, int cout, int ksize, int dtype, int out_mode, int cout_tile, float* dst_host);
/* Winograd form (rc_conv_desc.algo == 1): bytes of the packed U = G g G^T buffer (16 * cin * cout elements; 0 on an unsupported shape) and the host-side packer
 * (w_oihw: host fp32 (cout, cin, 3, 3); products accumulated in double, rounded once). */
size_t rc_wino_packed_bytes(int cin, int cout, int dtype);
int rc_wino_pack_weights(const float* w_oihw_host, int cin, int cout, int dtype, void* dst_host);
/* Number of partial-sum slots per image the conv kernel writes to chan_sums (4 waves per 8x32 tile; depends only on H,W). */
int rc_conv_sum_tiles(int height, int width);
/* Slots per image the launch described by `d` fills in chan_sums (pointers are only tested for NULL; d->chan_sums_slots is ignored): the answer comes from
 * the launcher itself, so an allocation sized by it cannot drift from the dispatch.  Either rc_conv_sum_tiles() (per (8x32 tile, wave)) or, for the
 * carried-sums kernels, grid x waves.  Consumers (rc_ca_gate, rc_ca_gate_ahead) fold whatever count they are given, in fixed order.  -1 on a bad desc. */
int rc_conv_sum_slots(const rc_conv_desc* d);
int rc_conv2d(const rc_conv_desc* desc, void* stream);
/* sizeof(rc_conv_desc) as compiled into the library: lets an FFI binding verify its struct mirror. */
size_t rc_conv_desc_size(void);

/* ---- a11 folded: the tail as ONE convolution ------------------------------------------------------
 * Replaces: self.tail = seq(conv(C, 4C, 'C'), nn.PixelShuffle(2), conv(C, 3, 'C')) (models/LiteISP.py:1996-2000, 2379-2383; applied at
 * :2033, :2410).  There is no activation between the two convolutions, so conv2(PixelShuffle(conv1(x))) is one linear map: a 5x5 convolution
 * C -> 4*O whose channel 4o + 2i + j is output channel o at sub-pixel (i, j) (each sub-pixel uses a 4x4 subset of the 5x5 taps; 36 % of the
 * folded weights are structurally zero).  It does 19 200 instead of 88 128 MACs per packed pixel and the 2H x 2W x C intermediate map (6.4 GB
 * written + 7.7 GB read per 8 frames of 4K) never exists.
 *   rc_tail_fold_weights (host): w1 (4C,C,3,3), b1 (4C) or NULL, w2 (O,C,3,3), b2 (O) or NULL -> wc (4O,C,5,5), bc (4O); accumulated in
 *     double.  Run the result as rc_conv2d ksize 5 (4O <= 16; bf16 C = 48 k / 32 k, fp32 C = 16 k) with RC_OUT_PIXEL_SHUFFLE2_NCHW.
 * The fold is exact everywhere except the outermost ring of output pixels (the second convolution zero-pads the shuffled map; the fold sees
 * conv1 evaluated beyond the edge there).  The ring is recomputed with the two original convolutions on four thin strips:
 *   rc_tail_ring_gather : x NHWC (B,H,W,C) -> rows (2B,2,W,C) = [x[:,0:2], x[:,H-2:H]], cols (2B,2,H,C) = [x[:,:,0:2], x[:,:,W-2:W]] TRANSPOSED
 *     (cols pixel (r, y) = x[b][y][r]: the side strips run as 2 x H images -- a 2-pixel-wide image would waste 15/16 of every conv tile)
 *   (caller: conv1 with RC_OUT_PIXEL_SHUFFLE2 + conv2 with RC_OUT_NCHW on both strip batches, the transposed batch with ky <-> kx swapped weights and
 *    conv1's sub-pixel order 4c+2i+j <-> 4c+2j+i -> rows_out (2B,O,4,2W), cols_out (2B,O,4,2H))
 *   rc_tail_ring_scatter: output row 0 / row 2H-1 from rows_out rows 0 / 3, column 0 / 2W-1 from cols_out ROWS 0 / 3 -> out (B,O,out_h,out_w),
 *     skipping what the crop (out_h < 2H, out_w < 2W) removes.  dtype = element type of the strips and of out. */
int rc_tail_fold_weights(const float* w1_host, const float* b1_host, const float* w2_host, const float* b2_host, int c, int o,
                         float* wc_host, float* bc_host);
int rc_tail_ring_gather(const void* d_x, void* d_rows, void* d_cols, int dtype, int batch, int H, int W, int c, void* stream);
int rc_tail_ring_scatter(const void* d_rows_out, const void* d_cols_out, void* d_out, int dtype, int batch, int c_out, int H, int W,
                         int out_h, int out_w, void* stream);

/* ---- a5+a6 fused: two 3x3 convolutions with the intermediate kept on chip ------------------------
 * Replaces, for 48-channel bf16 feature maps (the flagship width):
 *   RCABlock.res      conv -> ReLU -> conv                 (models/networks.py:296-311; + CALayer sums)
 *   Res_GFM           conv0 -> x*scale+shift+x -> LeakyReLU -> conv1 -> + x     (models/LiteISP.py:553-558)
 * i.e. exactly two rc_conv2d calls whose intermediate NHWC map never goes to HBM (the 48->48 layers run at
 * the HBM copy rate, so this halves their traffic).  Same operands as rc_conv_desc where they apply:
 * in1/in_gate/in_store feed conv1's input staging (x = in0*gate + in1), act1 is RC_ACT_RELU, or
 * RC_ACT_LEAKY together with film_scale/film_shift; residual is added to conv2's result (FiLM form only);
 * chan_sums receives rc_conv_pair_sum_slots() partials per image (ReLU form only) for rc_ca_gate.
 * w1/w2: rc_conv_pack_weights(48,48,3,RC_BF16,RC_OUT_NHWC); b1/b2: rc_conv_pack_bias or NULL.
 * Other widths / fp32 return RC_ERR_UNSUPPORTED-style errors: callers issue two rc_conv2d calls instead. */
typedef struct rc_conv_pair_desc {
    int32_t batch, height, width, channels;
    int32_t dtype;                  /* RC_BF16                                                    */
    const void* in0;                /* NHWC (B,H,W,48)                                            */
    const void* in1;                /* optional skip tensor (with in_gate)                        */
    const float* in_gate;           /* optional (B,48) fp32 CALayer gate                          */
    void* in_store;                 /* optional: materialised in0*gate + in1                      */
    const void* w1; const float* b1;
    const float* film_scale;        /* (B,48) fp32, FiLM form                                     */
    const float* film_shift;
    int32_t act1;                   /* RC_ACT_RELU | RC_ACT_LEAKY                                 */
    float act1_slope;
    const void* w2; const float* b2;
    const void* residual;           /* optional NHWC (B,H,W,48), added to conv2's result          */
    void* out;                      /* NHWC (B,H,W,48)                                            */
    float* chan_sums;               /* optional fp32 (B, rc_conv_pair_sum_slots(), 48)            */
} rc_conv_pair_desc;
int rc_conv_pair(const rc_conv_pair_desc* desc, void* stream);
int rc_conv_pair_sum_slots(int height, int width);
size_t rc_conv_pair_desc_size(void);

/* ---- a5 fused: the lens-shading MLP as one launch ------------------------------------------------
 * Replaces: Lens_Shading_Correction.model = Conv1x1(cin0,48) -> LeakyReLU -> [Conv1x1(48,48) -> LeakyReLU] x (n_mid-1)
 * -> Conv1x1(48,48) (models/LiteISP.py:363-378) when the width is 48 and the tensors are bf16: a 1x1 conv has no
 * halo, so each wave carries its 64 pixels through all layers in LDS and HBM sees only the coordinates in and the
 * final map out (layer by layer, every intermediate 48-channel map is written and re-read).
 * d_x NHWC (pixels, cin0 <= 8) bf16; d_w0packed = rc_conv_pack_weights(cin0,48,1,RC_BF16,RC_OUT_NHWC) (one zero-padded MFMA
 * step), d_b0 its packed bias (or NULL); d_wpacked[m] = rc_conv_pack_weights(48,48,1,RC_BF16,RC_OUT_NHWC) device buffers,
 * d_bias[m] packed fp32 (or NULL), 1 <= n_mid <= 4
 * (host arrays of device pointers); slope in [0,1]; d_out NHWC (pixels, 48) bf16.  fp32 / other widths: error,
 * callers run the layers through rc_conv2d. */
int rc_pointwise_chain48(const void* d_x, int cin0, const void* d_w0packed, const float* d_b0, const void* const* d_wpacked,
                         const float* const* d_bias, int n_mid, float slope, void* d_out, int dtype, long long pixels,
                         void* stream);

/* The chain with register-resident activations, optionally with the consuming convolution folded in (models/LiteISP.py:363-378 and
 * :2012-2014 `h = head(raw); h = h * (lsc(coord) + 1)`; ISPUNet :1352-1355; models/raw2bit.py:1775-1781 at width 128):
 *   d_out = chain(d_x)                                         when d_raw == NULL
 *   d_out = (conv3x3(d_raw) + bias) * (chain(d_x) + 1)         otherwise -- one launch, the lens-shading map never reaches HBM.
 * A wave carries 64 pixels through every layer in MFMA fragments (no LDS round trip between layers); the map is rounded to bf16 where the
 * two-launch path stores it.  bf16 only, c = 32, 48, 64 or 128, cin0 <= 4, raw_c <= 4, 1 <= n_mid <= 4 (layers after the first).
 * d_blob: rc_lsc_pack's output (rc_lsc_packed_bytes bytes) on the device: w0 (c,cin0), wmid[l] (c,c), whead (c,raw_c,3,3) fp32 as in the
 * state_dict (+ biases, or NULL) re-ordered into pair-packed MFMA fragments.  d_x NHWC (batch,H,W,cin0), d_raw NHWC (batch,H,W,raw_c). */
size_t rc_lsc_packed_bytes(int c, int n_mid, int has_head);
int rc_lsc_pack(const float* w0, const float* b0, int cin0, const float* const* wmid, const float* const* bmid, int n_mid,
                const float* whead, const float* bhead, int raw_c, int c, void* dst);
int rc_lsc_chain(const void* d_x, int cin0, const void* d_blob, int c, int n_mid, float slope, const void* d_raw, int raw_c,
                 void* d_out, int batch, int H, int W, void* stream);

/* ---- a8: CALayer gate -------------------------------------------------------------------------
 * Replaces: AdaptiveAvgPool2d(1) -> Conv1x1(C,C/r) -> ReLU -> Conv1x1(C/r,C) -> Sigmoid
 * (models/networks.py:259-269).  d_sums: (B, n_tiles, C) partials from rc_conv2d; w0 (Cr,C), b0 (Cr),
 * w1 (C,Cr), b1 (C) fp32 device; gate: (B,C) fp32.  Fixed-order reduction => run-to-run bitwise stable.
 * d_sums is scratch after the call (large images are folded in place in a first stage). */
int rc_ca_gate(float* d_sums, int batch, int n_tiles, int c, int cr, float inv_hw,
               const float* d_w0, const float* d_b0, const float* d_w1, const float* d_b1,
               float* d_gate, void* stream);

/* The same gate, computed AHEAD of the convolution whose output CALayer pools: RCABlock is x + CA(conv2(t)), t = relu(conv1(x))
 * (models/networks.py:296-311), and mean_HW(conv2(t)) is linear in t: b2 + 1/HW * sum_{c,tap} W2[o][c][tap] * S_tap[c], S_tap = the sum of t[c] over
 * the pixels tap (dy,dx) reaches inside the image (total - cut-off border row / column + corner).  d_sums: conv1's channel-sum partials of t
 * (B, n_tiles, C) (scratch after the call, as in rc_ca_gate); d_t: the NHWC map t itself (B,H,W,C), read for its four border lines only;
 * d_w2t (C_in,3,3,C_out) fp32 = conv2's OIHW weight permuted (1,2,3,0) / d_b2 (C) or NULL: conv2's parameters; d_scratch: rc_ca_gate_ahead_scratch_floats(B, C) floats.  The gate then goes
 * into conv2's launch as rc_conv_desc.out_scale (+ residual = x): one map written per RCAB body instead of r and r*gate + x. */
size_t rc_ca_gate_ahead_scratch_floats(int batch, int c);
int rc_ca_gate_ahead(float* d_sums, int batch, int n_tiles, int c, int cr, const void* d_t, int dtype, int H, int W,
                     const float* d_w2t, const float* d_b2, const float* d_w0, const float* d_b0, const float* d_w1, const float* d_b1,
                     float* d_scratch, float* d_gate, void* stream);

/* Per-channel partial sums of an NHWC map (B, n_pix, C): the AdaptiveAvgPool2d(1) of a standalone CALayer
 * (models/networks.py:259,268) when no producing conv emitted them.  d_sums: fp32 (B, rc_channel_sums_slots(n_pix), C),
 * fixed-order partials in the layout rc_ca_gate folds. */
int rc_channel_sums_slots(int n_pix);
int rc_channel_sums(const void* d_x, int dtype, int batch, int n_pix, int c, float* d_sums, void* stream);

/* y = r*gate[b][c] + x  (CALayer scale + RCAB skip, networks.py:270,311) for call sites where the
 * gated tensor is not consumed by a conv; d_x == NULL: y = r*gate (CALayer alone, networks.py:270).
 * NHWC, n_pix = H*W per image. */
int rc_gate_residual(const void* d_r, const float* d_gate, const void* d_x, void* d_y, int dtype,
                     int batch, int n_pix, int c, void* stream);

/* GFMLayer applied to a feature map (models/LiteISP.py:308-321; Res_GFM_LFM :601-620): y = x*scale[b][c] + shift[b][c] + x with the
 * two (B,C) fp32 vectors of rc_gfm_vector.  NHWC, n_pix = H*W per image. */
int rc_film_apply(const void* d_x, const float* d_scale, const float* d_shift, void* d_y, int dtype, int batch, int n_pix, int c, void* stream);

/* ---- a19 (SWAtten, models/tcm.py:284-289): y = a * sigmoid(b) + identity, element-wise on NHWC maps of n_elems
 * elements (a multiple of 16 bytes); y may alias a. */
int rc_sigmoid_gate_add(const void* d_a, const void* d_b, const void* d_identity, void* d_y, int dtype, long long n_elems,
                        void* stream);

/* ---- a20 (CompressAI layers under models/tcm.py:336-357, restated; parity unpinned) -----------------------------------
 * Stride-2 sampling of an NHWC map: dst (B, ceil(H/2), ceil(W/2), C)[y][x] = src (B,H,W,C)[2y][2x].  A 3x3 stride-2 padding-1
 * convolution (conv3x3(stride=2) in ResidualBlockWithStride / g_a / h_a) is rc_conv2d followed by this; a 1x1 stride-2
 * convolution (the block's skip) is this followed by rc_conv2d. */
int rc_subsample2(const void* d_src, void* d_dst, int dtype, int batch, int H, int W, int c, void* stream);
/* ---- a18 (models/raw2bit.py): nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) of HyCondModDecBlock (:790-793) on NHWC maps:
 * dst (B,2H,2W,c); source coordinate = dst * (H-1)/(2H-1) per axis. */
int rc_upsample_bilinear2(const void* d_src, void* d_dst, int dtype, int batch, int H, int W, int c, void* stream);
/* SpatialFeatureTransform (:877-885) + the block's identity (:313): y = x*scale + shift + x (+ identity if d_identity != NULL). */
int rc_sft_apply(const void* d_x, const void* d_scale, const void* d_shift, const void* d_identity, void* d_y, int dtype,
                 long long n_elems, void* stream);
/* Space-to-depth by 2 on NHWC maps of any width: dst (B, ceil(H/2), ceil(W/2), 4c)[y][x][(2i+j)*c + k] = src (B,H,W,c)[2y+i][2x+j][k], zero
 * beyond the bottom / right edge.  A 3x3 stride-2 padding-1 convolution is then a 3x3 stride-1 convolution over the 4c-channel map
 * with the taps re-indexed (row offset -1 <- phase 1 of the previous row pair, 0 <- phases 0 and 1), which rc_conv2d runs at the
 * OUTPUT resolution. */
int rc_space_to_depth2(const void* d_src, void* d_dst, int dtype, int batch, int H, int W, int c, void* stream);
/* nn.PixelShuffle(2) on NHWC maps of any width: dst (B,2H,2W,c)[2y+i][2x+j][k] = src (B,H,W,4c)[y][x][4k + 2i + j].  (rc_conv2d's
 * RC_OUT_PIXEL_SHUFFLE2 store covers c % 16 == 0; this is for the narrow tails, e.g. subpel_conv3x3(2N, 3, 2) of g_s.) */
int rc_pixel_shuffle2(const void* d_src, void* d_dst, int dtype, int batch, int H, int W, int c_out, void* stream);
/* The same shuffle written straight into NCHW: d_src (batch,H,W,4*c_out) NHWC -> d_dst (batch,c_out,2H,2W) -- the codecs' x_hat at the
 * module boundary (subpel_conv3x3(2N, 3, 2) closing g_s, models/tcm.py:364) without a separate layout pass. */
int rc_pixel_shuffle2_nchw(const void* d_src, void* d_dst, int dtype, int batch, int H, int W, int c_out, void* stream);
/* GDN / inverse GDN around a 1x1 rc_conv2d:  rc_square gives x^2 (the conv input, weights gamma, bias beta);
 * rc_gdn_apply gives y = x * rsqrt(norm) (inverse=0) or x * sqrt(norm) (inverse=1), + identity if d_identity != NULL
 * (the residual add that follows the GDN in both residual blocks). */
int rc_square(const void* d_x, void* d_y, int dtype, long long n_elems, void* stream);
int rc_gdn_apply(const void* d_x, const void* d_norm, const void* d_identity, void* d_y, int dtype, int inverse, long long n_elems,
                 void* stream);

/* ---- a19: the likelihood path TCM.forward asks of CompressAI's entropy models (models/tcm.py:441-446, 467-469, 475-477),
 * restated from their published definitions (parity unpinned); element-wise on NHWC maps, likelihoods always fp32.
 * rc_entropy_bottleneck: EntropyBottleneck.forward in eval mode + the z_hat of :443-446.
 *   d_params: (C, 58) fp32 per channel, the 1-3-3-3-3-1 cumulative-logit network with the parameter non-linearities already
 *   applied on the host: [softplus(_matrix0) 3 | _bias0 3 | tanh(_factor0) 3] then for layers 1..3
 *   [softplus(_matrix_l) 3x3 row-major (out, in) | _bias_l 3 | tanh(_factor_l) 3], then [softplus(_matrix4) 3 | _bias4 1];
 *   d_medians: (C) fp32 = quantiles[:, 0, 1].
 *   likelihood = max(|sigmoid(s*upper) - sigmoid(s*lower)|, bound), s = -sign(lower + upper), evaluated at round(z - med) + med -+ 0.5;
 *   z_hat = (round(t) - t + t) + med, t = z - med  (ste_round). */
int rc_entropy_bottleneck(const void* d_z, const float* d_params, const float* d_medians, void* d_z_hat, float* d_likelihood,
                          int dtype, long long n_pix, int channels, float likelihood_bound, void* stream);
/* GaussianConditional.forward(y, scales, means) in eval mode + the y_hat of :470:
 *   likelihood = max(Phi((0.5 - |q|)/s) - Phi((-0.5 - |q|)/s), bound), q = round(y - mu), s = max(scale, scale_bound (0.11)),
 *   Phi(x) = 0.5 erfc(-x / sqrt 2);  y_hat = ste_round(y - mu) + mu. */
int rc_gaussian_conditional(const void* d_y, const void* d_scale, const void* d_mu, void* d_y_hat, float* d_likelihood, int dtype,
                            long long n_elems, float scale_bound, float likelihood_bound, void* stream);
/* y_hat_slice += 0.5 * tanh(lrp)  (:478-479) */
int rc_tanh_half_add(const void* d_a, const void* d_lrp, void* d_out, int dtype, long long n_elems, void* stream);

/* ---- a10: Haar DWT / IDWT as the reference's frozen grouped conv -----------------------------
 * Replaces: DWTForward (models/networks.py:224-235) / DWTInverse (:238-249).  taps: device fp32
 * (4C,1,2,2) exactly as stored in the state_dict ("down1.3.weight", "up1.0.weight").
 * forward: (B,H,W,C) -> (B,H/2,W/2,4C), channel 4c+k.  inverse: (B,h,w,4C) -> (B,2h,2w,C).
 * taps_uniform != 0: the caller guarantees every channel carries the taps of channel 0 (true for the
 * reference's Haar init, networks.py:228-233); the kernel then keeps the 16 taps in scalar registers. */
int rc_dwt_forward(const void* d_x, void* d_y, const float* d_taps, int taps_uniform, int dtype,
                   int batch, int H, int W, int c, void* stream);
int rc_dwt_inverse(const void* d_x, void* d_y, const float* d_taps, int taps_uniform, int dtype,
                   int batch, int h, int w, int c4, void* stream);

/* ---- a6: Color_Condition_GFM (global colour prior) --------------------------------------------
 * Replaces: color_block (models/LiteISP.py:23-30) = Conv1x1 -> AvgPool2d(3,2,1,count_include_pad)
 * -> LeakyReLU(0.2) [-> InstanceNorm2d(affine)], and the closing Conv1x1 + AdaptiveAvgPool2d(1)
 * (models/LiteISP.py:345-361).  All fp32, NCHW (the cond image is small).
 *   rc_color_block:  y = lrelu_0.2(avgpool3s2p1(conv1x1(x)))   x:(B,cin,h,w) -> y:(B,cout,ho,wo)
 *                    with ho=(h-1)/2+1; if d_in_mean!=NULL the previous block's InstanceNorm
 *                    (x-mean)*rstd*gamma+beta is applied to x on load.
 *   rc_instance_stats: per (b,c) mean and 1/sqrt(var+eps) (biased var, eps 1e-5).
 *   rc_color_head:   v[b][o] = mean_hw(conv1x1(x))  -> (B,cout)   (the 5th block has no norm)      */
int rc_color_block(const void* d_x, int x_dtype, float* d_y, int batch, int cin, int cout, int h, int w,
                   const float* d_w, const float* d_b,
                   const float* d_in_mean, const float* d_in_rstd, const float* d_in_gamma,
                   const float* d_in_beta, void* stream);
int rc_instance_stats(const float* d_x, float* d_mean, float* d_rstd, int batch, int c, int hw,
                      float eps, void* stream);
/* InstanceNorm2d(affine) of a standalone CB block (models/LiteISP.py:215-230) with the statistics of rc_instance_stats:
 * y = (x - mean[b][c]) * rstd[b][c] * gamma[c] + beta[c], fp32 NCHW (inside the colour branch the norm is folded into the next
 * block's load instead). */
int rc_instance_norm(const float* d_x, float* d_y, const float* d_mean, const float* d_rstd, const float* d_gamma, const float* d_beta,
                     int batch, int c, int hw, void* stream);
int rc_color_head(const float* d_x, float* d_vec, int batch, int cin, int cout, int hw,
                  const float* d_w, const float* d_b, void* stream);

/* ---- a7: GFM vector MLPs ----------------------------------------------------------------------
 * Replaces: GFM_{scale,shift}_conv1(leaky_relu(GFM_{scale,shift}_conv0(vec), 0.1))
 * (models/LiteISP.py:554-555).  vec (B,cond_c); w0 (nf,cond_c) b0 (nf) w1 (c,nf) b1 (c); out (B,c). */
int rc_gfm_vector(const float* d_vec, int batch, int cond_c, int nf, int c,
                  const float* d_w0, const float* d_b0, const float* d_w1, const float* d_b1,
                  float* d_out, void* stream);

/* ---- a14-a16: GroupMix attention (GMA_Block) ---------------------------------------------------
 * Replaces the non-GEMM parts of models/groupmix.py:159-299 (identical copy models/raw2bit.py:98-142); the
 * four nn.Linear layers run through rc_conv2d as 1x1 convolutions (tokens (B,N,C) == NHWC pixels).
 *
 * rc_dwconv2d: depth-wise KxK (3/5/7), stride 1, 'same' zero padding, on channel sub-ranges of NHWC tensors:
 *   y[b,p, y_c0 + r*y_rep_stride + c] = bias[r*w_rep_stride + c] + sum_t wT[t][r*w_rep_stride + c] *
 *                                       x[b,p+t, x_c0 + r*x_rep_stride + c]  (+ x[b,p,..] if add_identity)
 *   for r < n_rep, c < n_ch.  wT: device fp32, TAP-MAJOR (K*K, n_w).  Replaces ConvPosEnc.proj
 *   (groupmix.py:206,215), SeparableConv2d.conv1 (:244) and ConvRelPosEnc.conv_list (:127-133).
 * rc_layernorm: nn.LayerNorm over C of (tokens, C) (groupmix.py:280,285).
 * rc_gma_pointwise: Aggregator tail (groupmix.py:92-100) with BatchNorm(eval) folded to scale/shift:
 *   qkv (B,N,3C), dw = depth-wise outputs of groups 1..3 (token t, rep r, group g at dw + t*dw_tok_stride +
 *   r*dw_rep_stride + (g-1)*seg), dwl = local depth-wise output (dwl + t*dwl_tok_stride + r*dwl_rep_stride); both may
 *   live in one (B,N,3,4seg) tensor written by a single rc_dwconv2d launch (strides 12seg / 4seg, dwl = dw + 3seg)
 *   -> qkvp (B,N,3,4seg) [q|k|v, channel = head*Ch + i], loc (B,N,seg).
 * rc_gma_kv: softmax over the N tokens fused with k^T v (groupmix.py:187-188): per-channel max pass, exp-sum +
 *   k^T v pass, fixed-order merge: ktv (B,heads,Ch,Ch) fp32 = scale * softmax_N(k)^T v.  d_scratch: rc_gma_kv_scratch_bytes().
 * rc_gma_apply: out (B,N,C) = [ q.ktv + q*convv | loc ]  (groupmix.py:189-194). */
/* ---- a15/a16 fused per-token stages of GMA_Block for dim 80 (5 x 16, 8 heads), bf16 (csrc/gma_fused.hip) --------------------
 * Chain weights: an nn.Linear / 1x1 conv weight (cout, cin) fp32 host -> bf16 MFMA fragments whose output rows are ordered so
 * that a layer's accumulator fragments are the next layer's B fragments (rc_chain_packed_bytes bytes); biases fp32 in the same
 * row order, zero padded (rc_chain_packed_rows(cout) floats; b NULL: zeros).  Host functions.
 * rc_gma_ln_qkv: qkv = Linear_qkv(LayerNorm1(x (tokens, 80))), written PLANAR BY 16-CHANNEL SEGMENT: d_qkv[15][tokens][16], segment
 *                s = channel / 16 = 5 * {q,k,v} + group -- the layout rc_gma_aggregate reads            (groupmix.py:178 after :293)
 * rc_gma_tail:   (d_qkvp, d_convv: the segment-planar tensors of rc_gma_aggregate / rc_gma_crpe; d_loc (tokens,16); d_x (tokens,80))
 *                y = [ q.ktv + q*convv | loc ] (groupmix.py:189-194); x2 = proj(y) + x (:197, :294); x3 = x2 + fc2(GELU(fc1(LN2(x2))))
 *                (:296-298); cout == 0: out (tokens, 80) = x3; cout == 192: out (tokens, 192) = Conv1x1(x3) + res (the cfg3 net's
 *                gma_out + d1, realcamnet_amd/LiteISP.py).  d_ktv (B,8,8,8) fp32 from rc_gma_kv; d_ktv_frags: B * 8 KiB scratch.
 * Rounding points are those of the layer-by-layer path (bf16 wherever that path stored a tensor), accumulation fp32; GELU's erf is
 * evaluated to 1.5e-7 absolute (far below the bf16 rounding of its result). */
size_t rc_chain_packed_bytes(int cin, int cout);
int rc_chain_packed_rows(int cout);
int rc_chain_pack_weights(const float* w, int cin, int cout, void* dst);
int rc_chain_pack_bias(const float* b, int cout, float* dst);
int rc_gma_ln_qkv(const void* d_x, void* d_qkv, long long tokens, const void* d_wpacked, const float* d_bias_packed,
                  const float* d_ln_gamma, const float* d_ln_beta, float eps, void* stream);
int rc_gma_tail(const void* d_qkvp, const void* d_convv, const void* d_loc, const void* d_x, const float* d_ktv, void* d_ktv_frags,
                int batch, int n_tok, const void* d_w_proj, const float* d_b_proj, const float* d_ln_gamma, const float* d_ln_beta,
                float eps, const void* d_w_fc1, const float* d_b_fc1, const void* d_w_fc2, const float* d_b_fc2, const void* d_res,
                const void* d_w_out, const float* d_b_out, int cout, void* d_out, void* stream);

/* rc_gma_aggregate: the whole Aggregator (groupmix.py:56-105) for dim 80, bf16, one launch: qkv in rc_gma_ln_qkv's segment-planar
 * layout [15][B,H,W][16] -> qkvp, segment-planar too: [12 = 4 * {q,k,v} + group][B,H,W][16]
 * [q|k|v][group 0: BN+Hardswish | groups 1..3: dw 3/5/7 -> pw 16x16 -> BN -> Hardswish] and loc (B,H,W,16) = Hardswish(LN(pw(dw3x3(
 * [q4|k4|v4])))).  dw3/5/7: tap-major (K*K,16) fp32; dwl (3,9,16); pw (3,16,16) [out][in]; pwl (16,48); bn_scale/shift (4,16)
 * (BatchNorm(eval) folded); ln (16).  Same values as rc_dwconv2d + rc_gma_pointwise up to the point-wise product's summation order. */
int rc_gma_aggregate(const void* d_qkv, void* d_qkvp, void* d_loc, int batch, int H, int W, const float* d_dw3, const float* d_dw5,
                     const float* d_dw7, const float* d_dwl, const float* d_pw, const float* d_pwl, const float* d_bn_scale,
                     const float* d_bn_shift, const float* d_ln_g, const float* d_ln_b, float* d_kmax, void* stream);

/* rc_gma_qkv_aggregate (ABI 13): rc_gma_ln_qkv + rc_gma_aggregate as ONE launch (groupmix.py:178 after :293, then :56-105) -- the 240-channel
 * qkv map is never written: a block LayerNorms x (B,H,W,80) for a 16 x 32 tile + halo once, keeps the normalised tokens in registers as MFMA B
 * fragments and produces qkv one 16-channel segment at a time into an LDS tile; the depth-wise K x K windows run ON THE MATRIX CORES as banded
 * Toeplitz products (one MFMA per channel, kernel row and 16-pixel group), then point-wise / BatchNorm / Hardswish as rc_gma_aggregate.
 * d_wq_natural: the (240, 80) qkv weight as rc_chain_pack_weights_natural fragments (output rows in natural channel order: 16-row tile m =
 * segment m); d_bq (240) fp32 or NULL; d_toeplitz: rc_gma_toeplitz_bytes() bytes from rc_gma_toeplitz_pack (host) over the same tap tensors
 * rc_gma_aggregate takes (dw3 / dw5 / dw7 tap-major (K*K,16), dwl (3,9,16)); the other arguments as rc_gma_ln_qkv / rc_gma_aggregate.
 * Same rounding points as the two-launch path; the depth-wise sums are formed in the matrix pipe's order (exact bf16 products, fp32 sums), so
 * results agree with it to the last fp32 bits before the bf16 rounding (tests hold: >= 99.9 % of the values bit-equal, the rest 1 bf16 ulp). */
int rc_chain_pack_weights_natural(const float* w, int cin, int cout, void* dst);   /* host; rc_chain_packed_bytes(cin, cout) bytes; 16 | cin, 16 | cout */
size_t rc_gma_toeplitz_bytes(void);
int rc_gma_toeplitz_pack(const float* dw3, const float* dw5, const float* dw7, const float* dwl, void* dst);   /* host */
/* rc_gma_in_cpe (ABI 14): the cfg3 net's `gma_in` (Conv1x1 192 -> 80, realcamnet_amd/LiteISP.py) followed by the block's ConvPosEnc (depth-wise 3x3 + identity,
 * groupmix.py:203-217, :293) as ONE launch: d_x (B,H,W,80) = a + dw3x3(a) + b_cpe, a = W_in d_d1 + b_in rounded to bf16 (zero outside the image: the depth-wise
 * convolution's padding); the 80-channel map a never reaches HBM.  bf16.  d_w_in_natural: rc_chain_pack_weights_natural(192 -> 80) fragments; d_b_in / d_b_cpe (80)
 * fp32 or NULL; d_toeplitz3: 3 * 80 KiB from rc_dw_toeplitz_pack(taps, 3, 80, .) (host; taps tap-major (9, 80) fp32).  The depth-wise sums are formed in the
 * matrix pipe's order and the 1x1 sums add the bias last: equal to rc_conv2d + rc_dwconv2d on > 99.8 % of the values, the rest within a few bf16 ulps of the
 * intermediate (tested); bitwise stable from run to run (tested). */
int rc_dw_toeplitz_pack(const float* taps, int K, int n_ch, void* dst);        /* host; K * n_ch KiB; K = 3, 5 or 7 */
int rc_gma_in_cpe(const void* d_d1, const void* d_w_in_natural, const float* d_b_in, const void* d_toeplitz3, const float* d_b_cpe, void* d_x,
                  int batch, int H, int W, void* stream);
int rc_gma_qkv_aggregate(const void* d_x, const void* d_wq_natural, const float* d_bq, const float* d_ln1_gamma, const float* d_ln1_beta, float eps,
                         void* d_qkvp, void* d_loc, int batch, int H, int W, const void* d_toeplitz, const float* d_pw, const float* d_pwl,
                         const float* d_bn_scale, const float* d_bn_shift, const float* d_ln_g, const float* d_ln_b, float* d_kmax, void* stream);

/* d_kmax (optional, may be NULL): (batch, 64) fp32, on return the per-channel maximum over the image of the aggregated k (the stored bf16
 * values) -- the shift of softmax_N(k) (models/groupmix.py:190) -- accumulated by the aggregator itself with integer atomics (a maximum is
 * order-independent, so this stays bitwise reproducible).  With it the two-pass rc_gma_kv_planar is replaced by ONE pass on the matrix cores:
 *   rc_gma_kv_mfma: ktv[b][h][i][j] = scale * sum_t softmax_t(k)[t][h,i] v[t][h,j] for 8 heads x 8 channels, segment-planar bf16 qkv'
 * (d_scratch: rc_gma_kv_mfma_scratch_bytes bytes).  exp(k - max) is rounded to bf16 for the MFMA (numerator and denominator use the same
 * rounded values); accumulation fp32, block partials merged in fixed order. */
/* Transformer-block MLP of the codecs (models/tcm.py:234-235: x + mlp(ln2(x)), mlp = Linear(C,4C) -> GELU -> Linear(4C,C)) as ONE launch with
 * register-resident activations: d_out = d_x + fc2(GELU(fc1(LayerNorm(d_x)))).  bf16, token width c = 32 or 64, d_x / d_out (tokens, c);
 * weights packed by rc_chain_pack_weights(c -> 4c) and (4c -> c), biases by rc_chain_pack_bias (or NULL).  Rounding points are those of
 * rc_layernorm + two rc_conv2d launches (LayerNorm output and GELU output to bf16). */
int rc_ln_mlp(const void* d_x, void* d_out, long long tokens, int c, const void* d_w_fc1, const float* d_b_fc1, const void* d_w_fc2,
              const float* d_b_fc2, const float* d_ln_gamma, const float* d_ln_beta, float eps, void* stream);

/* LayerNorm + Linear in one launch (the W-MSA embedding layer behind ln1, models/tcm.py:179-181, 232): d_out (tokens, cout) =
 * Linear(LayerNorm(d_x (tokens, c))), bf16, c = 32 or 64, cout a multiple of 32 (<= 512); weights by rc_chain_pack_weights(c -> cout). */
int rc_ln_linear(const void* d_x, void* d_out, long long tokens, int c, int cout, const void* d_w, const float* d_b, const float* d_ln_gamma,
                 const float* d_ln_beta, float eps, void* stream);
/* (ABI 14) the same launch with d_out SEGMENT-PLANAR: [cout / 8 segments][tokens][8 channels] -- the q / k / v layout rc_window_attention_planar8 reads. */
int rc_ln_linear_planar8(const void* d_x, void* d_out, long long tokens, int c, int cout, const void* d_w, const float* d_b, const float* d_ln_gamma,
                         const float* d_ln_beta, float eps, void* stream);

/* GDN / inverse GDN (compressai.layers.GDN inside ResidualBlockWithStride / ResidualBlockUpsample; call sites models/tcm.py:336-364) as one
 * per-token launch: d_out = d_x * rsqrt(beta + gamma . d_x^2)  (inverse != 0: * sqrt(..))  [+ d_identity], bf16, c = 64 or 128 channels,
 * tensors (tokens, c).  d_gamma_packed = rc_chain_pack_weights of the EFFECTIVE (re-parametrised) gamma (c, c), d_beta_packed =
 * rc_chain_pack_bias of the effective beta.  Same rounding points as rc_square -> rc_conv2d -> rc_gdn_apply. */
int rc_gdn_chain(const void* d_x, const void* d_identity, void* d_out, long long tokens, int c, const void* d_gamma_packed,
                 const float* d_beta_packed, int inverse, void* stream);

/* Linear over the channel concatenation of two token maps + residual, without the concatenated map (the closing
 * `conv1_2(torch.cat((conv_x, trans_x), dim=1)) + x` of ConvTransBlock, models/tcm.py:265-267; raw2bit.py:324-327):
 * d_out (tokens, c) = d_residual + W . [d_a (tokens, c/2) ; d_b (tokens, c/2)] + bias.  bf16, c = 64 or 128; d_w = rc_chain_pack_weights(c -> c),
 * d_bias = rc_chain_pack_bias (or NULL), d_residual may be NULL; d_a_add (or NULL): the first half is d_a + d_a_add, rounded to bf16
 * (ConvTransBlock's `conv_block(conv_x) + conv_x`, models/tcm.py:262, without its own launch). */
int rc_cat_linear(const void* d_a, const void* d_a_add, const void* d_b, const void* d_residual, void* d_out, long long tokens, int c,
                  const void* d_w, const float* d_bias, void* stream);

int rc_gma_kv_mfma_blocks(int n_tok);
size_t rc_gma_kv_mfma_scratch_bytes(int batch, int n_tok);
int rc_gma_kv_mfma(const void* d_qkvp, int batch, int n_tok, float scale, const float* d_kmax, float* d_scratch, float* d_ktv, void* stream);

/* rc_gma_crpe: ConvRelPosEnc's depth-wise conv of v (groupmix.py:127-133,146-150) for dim 80 / 8 heads of 8, bf16: v = segments 8..11
 * of rc_gma_aggregate's segment-planar qkvp -> convv, segment-planar [4][B,H,W][16]; four 16-channel segments with windows 3, 5, 7, 7 (tap-major (K*K,16) fp32 each; segment 2 holds
 * heads of window 5 and 7, its window-5 taps zero-padded to 7x7), bias (64).  Same values as rc_dwconv2d. */
int rc_gma_crpe(const void* d_qkvp, void* d_convv, int batch, int H, int W, const float* d_taps0, const float* d_taps1,
                const float* d_taps2, const float* d_taps3, const float* d_bias, void* stream);

int rc_dwconv2d(const void* d_x, int x_stride_c, int x_c0, void* d_y, int y_stride_c, int y_c0, int dtype,
                int batch, int H, int W, int n_ch, int ksize, const float* d_wT, int n_w, const float* d_bias,
                int n_rep, int x_rep_stride, int y_rep_stride, int w_rep_stride, int add_identity,
                const int* d_kvec /* optional: true window per 16-byte weight vector when taps are zero-padded to K */,
                void* stream);
int rc_layernorm(const void* d_x, void* d_y, int dtype, long long tokens, int c, const float* d_gamma,
                 const float* d_beta, float eps, void* stream);
int rc_gma_pointwise(const void* d_qkv, const void* d_dw, const void* d_dwl, int dw_tok_stride, int dw_rep_stride,
                     int dwl_tok_stride, int dwl_rep_stride, void* d_qkvp, void* d_loc, int dtype, long long tokens, int c,
                     const float* d_pw, const float* d_bn_scale, const float* d_bn_shift, const float* d_pwl,
                     const float* d_ln_g, const float* d_ln_b, void* stream);
int rc_gma_kv_blocks(int n_tok);
size_t rc_gma_kv_scratch_bytes(int batch, int n_tok, int heads, int ch);
int rc_gma_kv(const void* d_qkvp, int dtype, int batch, int n_tok, int heads, int ch, float scale, float* d_scratch,
              float* d_ktv, void* stream);
/* rc_gma_kv on rc_gma_aggregate's segment-planar qkvp ([12][B * n_tok][16], bf16). */
int rc_gma_kv_planar(const void* d_qkvp, int batch, int n_tok, int heads, int ch, float scale, float* d_scratch, float* d_ktv, void* stream);
int rc_gma_apply(const void* d_qkvp, const void* d_convv, const void* d_loc, const float* d_ktv, void* d_out, int dtype,
                 int batch, int n_tok, int heads, int ch, int seg, void* stream);

/* ---- a18 plumbing: channel split / concat of NHWC tensors -----------------------------------------
 * Replaces torch.split / torch.cat along the channel dim in ConvTransBlock.forward (models/tcm.py:261-266):
 * dst[p, dst_c0 + c] = src[p, src_c0 + c] for c < n_ch, all offsets / counts whole 16-byte vectors. */
int rc_channel_copy(const void* d_src, int src_stride_c, int src_c0, void* d_dst, int dst_stride_c, int dst_c0, int n_ch,
                    long long pixels, int dtype, void* stream);
/* torch.cat of 1..8 NHWC tensors along the channel dim in ONE launch (the slice loop's `torch.cat([latent_means] + support_slices, dim=1)`,
 * models/tcm.py:460-466, 609-617: 2..6 parts, 22 times per forward): d_parts[k] is (pixels, widths[k]) dense, dst (pixels, sum widths). */
int rc_channel_concat(const void* const* d_parts, const int* widths, int n_parts, void* d_dst, long long pixels, int dtype, void* stream);

/* ---- a17: TCM window attention -------------------------------------------------------------------
 * Replaces the core of WMSA.forward (models/tcm.py:179-206): per ws x ws window of the cyclically shifted NHWC map
 * and per head, softmax(q k^T / sqrt(hd) + relpos[h, dy, dx] (+ -inf across the wrap in the last window row/column
 * when shifted)) v.  d_qkv (B,H,W,3C) = embedding_layer output on the UN-shifted map ([q | k | v], head-major inside
 * each), d_relpos = relative_position_params (heads, 2ws-1, 2ws-1) fp32, d_out (B,H,W,C) attention output at the
 * un-shifted pixel positions (ready for `linear`).  shift = 0 for type 'W', ws/2 for 'SW'.  The Linear layers,
 * LayerNorms and the MLP of tcm.Block (:214-236) are rc_conv2d (1x1) / rc_layernorm. */
int rc_window_attention(const void* d_qkv, const float* d_relpos, void* d_out, int dtype, int batch, int H, int W, int C,
                        int head_dim, int window, int shift, void* stream);
/* (ABI 14) the same attention over a SEGMENT-PLANAR d_qkv: [3C / 8 segments][B H W pixels][8 channels], segment = channel / 8 of the (B,H,W,3C) form above
 * (rc_ln_linear_planar8 writes it).  With head_dim 8 a lane of the interleaved form reads 16 of a pixel record's 384 bytes -- one 64-byte sector per lane; in its
 * segment's plane the 8 pixels of a window row are 128 contiguous bytes.  Exists for the matrix-core form only (bf16, 8 x 8 windows, B H W 3C < 2^31:
 * rc_window_attention_planar8_ok() != 0); RC_ERR_UNSUPPORTED otherwise.  Same arithmetic, same bits as rc_window_attention on the interleaved tensor. */
int rc_window_attention_planar8(const void* d_qkv, const float* d_relpos, void* d_out, int dtype, int batch, int H, int W, int C,
                                int head_dim, int window, int shift, void* stream);
int rc_window_attention_planar8_ok(int dtype, int batch, int H, int W, int C, int window);

/* ---- measurement helpers (bench.py): HIP-event timing of every rc_conv2d launch on its stream -
 * rc_prof_enable(1) brackets each subsequent rc_conv2d with hipEventRecord on the launch stream;
 * rc_prof_collect() synchronises the events and returns launches / total ms / total algorithmic
 * FLOPs (2*MAC at the padded size) since the last rc_prof_enable(1). */
/* A/B switches for tests and benches.  "persist": 0 = general kernel only, 1 (default) = automatic choice between the
 * persistent weights-resident kernel and its producer/consumer (wave-specialised) form, 2 = producer/consumer
 * wherever eligible, 3 = persistent only.  Results must be identical in all modes.
 * Others (all bit-identical alternatives of one operation, defaults in brackets): "conv32" [0] which layers take the
 * 32x32x16 MFMA forms (0 none, 4 = where measured faster, 1 / 2 / 3 = everywhere eligible in one of three forms; set BEFORE
 * packing: the packed weight order depends on it -- and on no other knob: such a layer runs its own kernel in every "persist" mode); "pair_impl" [0] which of the two rc_conv_pair kernels; "pss" [0] the
 * 48 -> 192 + PixelShuffle layer with its output staged through LDS; "dw3_seg16" [1] rc_dwconv2d's bf16 3x3 single-rep
 * case on 16-channel segments; "conv_flags" knock-outs for timing experiments (1 no stores, 8 stores into one 4 MB
 * window: results are then meaningless; 64 one persistent block per CU instead of two: results unchanged, an occupancy experiment);
 * "persist_auto" [1] the wave-autonomous kernels 6 / 7 (0 off, 1 all but the residual forms of kernel 6, 2 every eligible form); "sums_compact" [1] the compact
 * channel-sum slot layout (rc_conv_sum_slots); "thin" [2] kernel 4b (one barrier per stage) in place of the multi-chunk kernel 4: 1 = only where the stages are thin (weights and tiles by LDS-DMA, tiles two stages
 * ahead), 2 = also the 3x3 layers with <= 36 KB of weights per chunk (tile through registers, one stage ahead), 0 = kernel 4; "lds_poison" [0] test aid: every rc_conv2d launch is preceded by rc_debug_poison_lds (NaNs in all LDS). "wino_nnt" [0] the Winograd kernel's item size: 0 automatic (4 x 16-pixel items where one image's 4 x 32 regions quantise badly over the chip, 4 x 32 otherwise), 1 / 2 forced. */
int rc_debug_set(const char* key, int value);
/* Current value of an integer knob ("persist", "conv32", "conv_flags", "pss"), -1 for an unknown key.  The host mirror keys its packed-weight cache
 * on "conv32" (the packed order of the 32x32x16 layers depends on it and on nothing else). */
int rc_debug_get(const char* key);
/* "conv_phase_timing": device buffer of >= 512 int64; the producer/consumer conv kernel then records s_memtime
 * cycle counts per tile phase for one compute wave and one loader wave (NULL switches it off). */
int rc_debug_set_ptr(const char* key, void* d_ptr);
/* Calibration: sustained rate of back-to-back v_mfma_f32_16x16x32_bf16 on every SIMD (1 or 2 waves per SIMD),
 * and s_memtime ticks per MFMA per SIMD: what this part's clocks allow, to read MFMA utilisation against. */
/* Experiment: a HIP stream confined to half of the chip's CUs (hipExtStreamCreateWithCUMask); kind 0/1 = lower / upper 128
 * mask bits, 2/3 = lower / upper 16 bits of every 32-bit word.  Never destroyed (debug only). */
int rc_debug_stream_create_masked(int kind, void** stream_out);
/* HBM streaming probe on the default stream: mode 0 copy / 1 read / 2 write of `bytes`; nt = non-temporal accesses;
 * contiguous = one range per block (else grid-stride); blocks = 0 -> one-shot grid (4 x 16 B per thread). */
int rc_debug_hbm_probe(const void* src, void* dst, size_t bytes, int mode, int nt, int contiguous, int blocks, int iters,
                       double* ms_per_iter);
/* waves_per_simd 11 / 12 (both calibrations): the same loop on RANDOM operands that change from MFMA to MFMA (4 A x 4 B register sets per wave):
 * on toggling data the part runs at its board power cap, not at its maximum clock (tools/power_probe.py); ticks are not reported (0). */
int rc_debug_mfma_peak(int waves_per_simd, int iters, double* tflops, double* memtime_ticks_per_mfma);
/* The same calibration on v_mfma_f32_32x32x16_bf16 (8 independent accumulators per wave). */
int rc_debug_mfma_peak32(int waves_per_simd, int iters, double* tflops, double* memtime_ticks_per_mfma);
/* Test aid: fill the whole LDS allocation of every CU with `pattern` (e.g. 0x7FC07FC0: bf16 / fp32 NaNs) on `stream`, so that a kernel reading LDS it has not
 * written, or before the write has landed, fails the parity tests whatever ran before it (LDS contents survive from one kernel to the next). */
int rc_debug_poison_lds(unsigned pattern, void* stream);
int rc_prof_enable(int on);
int rc_prof_collect(int64_t* n_launches, double* total_ms, double* total_flops);
/* The same records grouped by layer shape (cin, cout, ksize): launches, summed HIP-event ms, algorithmic FLOPs and algorithmic BYTES (every map
 * the launch has to touch once: input, output, residual / multiplier operand, the gated form's skip and materialised input) -- what bench.py's
 * roofline line names its dominant kernel from.  Does not clear the records (rc_prof_enable does). */
typedef struct rc_prof_row { int cin, cout, ksize, launches; double ms, flops, bytes; } rc_prof_row;
int rc_prof_collect_rows(rc_prof_row* rows, int max_rows, int* n_rows);

/* ---- f3: on-device entropy coding (csrc/rans.hip) ----------------------------------------------------------
 * Replaces the `.tolist()` symbol dumps + CompressAI `BufferedRansEncoder` / `RansDecoder` calls of compress() / decompress()
 * (models/tcm.py:511-570, 592-637; models/raw2bit.py:1876-1944, 1961-2027) and the table construction of `update()`
 * (models/tcm.py:430-435).  The coder is CompressAI's published rANS (64-bit state, 32-bit words, 16-bit probabilities, 4-bit bypass
 * escapes); CompressAI itself is absent from /root/reference: restated, parity unpinned.
 *
 * rc_pmf_to_quantized_cdf (host): `compressai._CXX.pmf_to_quantized_cdf`: n probabilities -> n + 1 cumulative frequencies.
 * rc_gc_symbols: GaussianConditional side of compress(): y, mu, scale NHWC (B, hw, C) -> symbols = round(y - mu) and CDF indexes
 *   (`build_indexes`: scale table search with the 0.11 lower bound) as int32 in the coder's (b, c, hw) order, y_hat = symbols + mu
 *   (NHWC).  d_y == NULL: indexes only (decompress()).   rc_gc_dequantize: y_hat = symbols + mu.
 * rc_eb_symbols: EntropyBottleneck side: symbols = round(z - median[c]), index = c, z_hat (encode = 1), or z_hat from symbols (0).
 * rc_rans_encode_chunks: two launches -- all symbols' (CDF row, escape, start, freq, reciprocal of freq) in parallel into d_scratch
 *   (rc_rans_encode_scratch_bytes, 8-byte aligned), then one lane per chunk of `chunk` consecutive symbols doing only the state updates
 *   (division-free, bit-identical to the dividing form: rc_debug_rans_rcp_selftest); every chunk a complete stream in BufferedRansEncoder's
 *   layout written at the END of its rc_rans_chunk_words(chunk)-word slot of d_words; d_nbytes[chunk] = its length (-1: bad index).
 *   rc_rans_compact gathers the streams at the given byte offsets.   rc_rans_decode_chunks: the inverse; every chunk's stream is bounded by the
 *   next chunk's offset (the last by stream_bytes): d_err = 1 bad CDF index, 2 truncated / corrupt stream (a read past a chunk's end, a chunk
 *   shorter than the flushed state or not word-sized, an escape longer than 8 nibbles) -- never an out-of-bounds read.
 * rc_rans_encode_host / rc_rans_decode_host: the same primitives on the host for ONE stream over all symbols = CompressAI's wire
 *   format; decode keeps the decoder state in state[2] (zero-initialised at the start of a stream) like RansDecoder.decode_stream. */
int rc_pmf_to_quantized_cdf(const float* pmf, int n, int precision, int32_t* cdf);
int rc_gc_symbols(const void* d_y, const void* d_mu, const void* d_scale, int dtype, int batch, long long hw, int channels,
                  const float* d_scale_table, int n_levels, float scale_bound, int32_t* d_symbols, int32_t* d_indexes, void* d_y_hat,
                  void* stream);
int rc_gc_dequantize(const int32_t* d_symbols, const void* d_mu, int dtype, int batch, long long hw, int channels, void* d_y_hat, void* stream);
int rc_eb_symbols(const void* d_z, const float* d_medians, int dtype, int batch, long long hw, int channels, int encode, int32_t* d_symbols,
                  int32_t* d_indexes, void* d_z_hat, void* stream);
int rc_rans_chunk_words(int chunk);
size_t rc_rans_encode_scratch_bytes(long long n, int chunk);
int rc_rans_encode_chunks(const int32_t* d_symbols, const int32_t* d_indexes, long long n, int chunk, const int32_t* d_cdf, int cdf_stride,
                          int n_cdfs, const int32_t* d_cdf_sizes, const int32_t* d_offsets, uint32_t* d_words, int32_t* d_nbytes, void* d_scratch,
                          void* stream);
long long rc_debug_rans_rcp_selftest(long long trials, unsigned long long seed);
int rc_rans_compact(const uint32_t* d_words, int chunk, const int32_t* d_nbytes, const long long* d_offsets, long long n_chunks, void* d_out,
                    void* stream);
int rc_rans_decode_chunks(const void* d_stream, long long stream_bytes, const long long* d_offsets, const int32_t* d_indexes, long long n, int chunk,
                          const int32_t* d_cdf, int cdf_stride, int n_cdfs, const int32_t* d_cdf_sizes, const int32_t* d_cdf_offsets, int32_t* d_symbols,
                          int32_t* d_err, void* stream);
long long rc_rans_encode_host(const int32_t* symbols, const int32_t* indexes, long long n, const int32_t* cdf, int cdf_stride, int n_cdfs,
                              const int32_t* cdf_sizes, const int32_t* offsets, void* out, long long out_cap);
int rc_rans_decode_host(const void* stream_bytes, long long n_bytes, unsigned long long* state, const int32_t* indexes, long long n, const int32_t* cdf,
                        int cdf_stride, int n_cdfs, const int32_t* cdf_sizes, const int32_t* offsets, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* REALCAM_HIP_H */
