/* tvae_hip.h -- C ABI of libtvae_hip.so: the MI355X (gfx950) kernels of the TARGET-VAE training hot path.
 *
 * The reference (SMLC-NYSBC/TARGET-VAE) is pure Python on PyTorch and has no FFI of its own; every entry
 * point below replaces a group of ATen call sites of the reference hot path (file:line cited per function,
 * paths relative to the reference root).  Conventions:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer to fp32 (or int32 where stated);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing allocates, frees or
 *     synchronises (graph-capturable); workspaces are provided by the caller;
 *   - return value: 0 on success, otherwise the hipError_t of the failed launch;
 *   - activations are FEATURE-MAJOR: X[feature][column] with column = image*positions + position
 *     contiguous (ld = leading dimension in floats).  "act" codes: 0 none, 1 LeakyReLU(slope), 2 tanh.
 */
#ifndef TVAE_HIP_H
#define TVAE_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef void* tvae_stream_t;

int tvae_abi_version(void);

/* The library keeps NO process-wide state.  The arithmetic of a matrix product is chosen per call by the entry point:
 * tvae_conv1_fwd / tvae_conv1_wgrad / tvae_linear_* compute exact fp32 products (v_mfma_f32_32x32x2_f32); the *_x6 and
 * *_dft entry points compute in the "x6" arithmetic (every fp32 operand split EXACTLY into three bf16 numbers, six
 * v_mfma_f32_32x32x16_bf16 partial products, fp32 accumulate: fp32-equivalent results) when called with parts = 3, in the
 * "h3" arithmetic (two fp16 parts under power-of-two scales, three v_mfma_f32_32x32x16_f16 products: fp32-equivalent
 * results with half the matrix instructions; tvae_dense_split2h below) with parts = 2, and with operands rounded to one
 * bf16 number (throughput mode, NOT fp32-equivalent) with parts = 1.  tvae_abi_version() == 6.  ABI 4: parts = 2,
 * tvae_dense_split2h; the buffers sized by tvae_conv1_dft_at_floats and tvae_linear_wgrad_x6_ws_floats carry extra words
 * -- the operand maxima of the h3 arithmetic -- behind their data.  ABI 5: the h3 scale is ONE POWER OF TWO PER ROW of an
 * operand in the sense of the product (a row / column of the output), not one per tensor: tvae_dense_split2h keeps one
 * maximum per row; tvae_conv1_fwd_dft leaves per-(frequency, image), per-(frequency, filter row), per-filter-row and
 * per-channel maxima behind A^T (the LAST C floats = max |out| per channel) and `amax_a1` of the encoder-tail entry points
 * points to C words.  An output row whose operand row lies 2^-32 below the rest of its tensor is as accurate, relative to
 * itself, as any other (tests/test_hip_primitives.py::test_h3_row_dynamic_range: 1e-5 per row against fp64). */

/* ---- rotated filter bank: GroupConv.trans_filter, src/models.py:174-197 (F.affine_grid + F.grid_sample x R) ----
 * weight [C][Cin][k*k] -> bank [(c*R + r)][ci*k*k + d].  tap_idx/tap_w [R][k*k][4]: bilinear taps of the fixed
 * rotations (idx < 0 = outside, zero padding).  bwd applies the transposed operator through a CSR table
 * (csr_ptr [k*k+1], entries r / dst / w), writing (or accumulating into) dweight. */
int tvae_rotate_bank_fwd(const float* weight, const int* tap_idx, const float* tap_w, float* bank, int C, int Cin,
                         int ksz, int R, tvae_stream_t stream);
int tvae_rotate_bank_bwd(const float* dbank, const int* csr_ptr, const int* csr_r, const int* csr_dst,
                         const float* csr_w, float* dweight, int C, int Cin, int ksz, int R, int accumulate,
                         tvae_stream_t stream);

/* ---- lifting convolution: GroupConv.forward, src/models.py:202-225 (F.conv2d + bias) fused with the following
 * activation (models.py:355).  y [B][Cin][n][n]; bank [C*R][Cin*k*k]; bias [C] (may be NULL);
 * out feature-major [C][B][R][Ho*Ho] (ld = B*R*Ho*Ho), Ho = n + 2*pad - k + 1, stride 1. */
int tvae_conv1_fwd(const float* y, const float* bank, const float* bias, float* out, int B, int Cin, int n, int ksz,
                   int pad, int C, int R, int act, float slope, tvae_stream_t stream);
/* weight gradient of the same convolution (autograd of F.conv2d, models.py:215): dbank [C*R][Cin*k*k] =
 * sum over (img, position) of dpre[c][img][r][p] * window(y).  dpre is the PRE-activation gradient in the layout
 * of `out`.  ws: split-K workspace of ws_floats floats (>= 2*C*R*Cin*k*k recommended; fewer disables split-K). */
int tvae_conv1_wgrad(const float* y, const float* dpre, float* dbank, float* ws, long ws_floats, int B, int Cin,
                     int n, int ksz, int pad, int C, int R, tvae_stream_t stream);

/* ---- dense layers (nn.Conv3d(.,.,1) src/models.py:347-351,356-358,390-392; nn.Linear src/models.py:107-121) ----
 * fwd:   Y[M][N] = act( W[M][K] X[K][N] + bias[m] + gbias[(n/group)][m] + res[m][n] )   (NULL terms skipped)
 * dgrad: dX[K][N] = ( W^T dpre[M][N] + add[K][N] ) * act'(aux[K][N])   (mask code as act; aux = saved output)
 * wgrad: dW[M][K] (+)= sum_n dpre[M][n] X[K][n]      (split-K through ws) */
int tvae_linear_fwd(const float* W, const float* X, const float* bias, const float* gbias, int group,
                    const float* res, float* Y, int M, int N, int K, long ldx, long ldy, int act, float slope,
                    tvae_stream_t stream);
int tvae_linear_dgrad(const float* W, const float* dpre, const float* add, const float* aux, float* dX, int M, int N,
                      int K, long ldd, long ldx, int mask, float slope, tvae_stream_t stream);
int tvae_linear_wgrad(const float* dpre, const float* X, float* dW, float* ws, long ws_floats, int M, int N, int K,
                      long ldd, long ldx, int accumulate, tvae_stream_t stream);

/* ---- reductions / skinny products used by bias grads, coordinate layer, last decoder layer ----
 * rowdot_seg: out[seg][m][o] = sum_{n in seg} X[m][n] * V[n][o]   (V NULL -> ones, no = 1; no in {1,2,3,4});
 *             amax (optional, ABI 7): one ZEROED device word that also receives max |X| (the h3 bound of X's consumer)
 * seg_sum:    out[i] (+)= scale * sum_s in[s][i]
 * coldot:     out[n][o] = b[o] + sum_m W[m*wsm + o*wso] X[m][n]
 * outer_mask: D[m][n] = (sum_o W[m*wsm + o*wso] dy[n][o]) * act'(H[m][n])
 * act_bwd:    dpre[i] = dY[i] * act'(Y[i]) */
int tvae_rowdot_seg(const float* X, long ldx, const float* V, int no, int M, int N, int seglen, float* out, float* amax,
                    tvae_stream_t stream);
int tvae_seg_sum(const float* in, int S, long L, float* out, float scale, int accumulate, tvae_stream_t stream);
int tvae_coldot(const float* X, long ldx, int M, int N, const float* W, int wsm, int wso, const float* bias, int no,
                float* out, tvae_stream_t stream);
int tvae_outer_mask(const float* dy, int no, const float* W, int wsm, int wso, const float* H, long ldh, float* D,
                    long ldd, int M, int N, int act, float slope, tvae_stream_t stream);
int tvae_act_bwd(const float* dY, const float* Y, float* dpre, long n, int act, float slope, tvae_stream_t stream);

/* ---- lifting convolution on the bf16 matrix pipe with fp32-equivalent results ("x6") ------------------------------
 * Same operator as tvae_conv1_fwd / tvae_conv1_wgrad (F.conv2d in GroupConv.forward, src/models.py:215, and its
 * weight gradient), evaluated with every fp32 operand split EXACTLY into three bf16 numbers and six bf16 MFMAs per
 * product block (fp32 accumulate; the dropped cross terms are < 2^-23 relative, below fp32 FMA-chain rounding).
 * 2.67x the matrix rate of the fp32 MFMA.  The operands are pre-split into 16-byte k-octet cells:
 *   tvae_bank_split3     bank [C*R][Cin*ksz*ksz] -> a3 (tvae_conv1_x6_bank_bytes bytes), once per step;
 *   tvae_conv1_fwd_x6    same arguments as tvae_conv1_fwd with a3 instead of bank;
 *   tvae_dy_split3       dY [C][B*R*Ho*Ho] (the dpre of tvae_conv1_wgrad) -> d3 (tvae_conv1_x6_dy_bytes bytes);
 *   tvae_conv1_wgrad_x6  same as tvae_conv1_wgrad with d3 instead of dpre; ws must hold at least C*R*Cin*ksz*ksz
 *                        floats (one partial slab).
 * tvae_conv1_x6_supported: 1 when both kernels fit the 160 KiB LDS for this geometry (else use the fp32 entry
 * points); the three query functions are pure host functions (no stream, no GPU work). */
int tvae_conv1_x6_supported(int Cin, int n, int ksz, int pad);
long tvae_conv1_x6_bank_bytes(int C, int R, int Cin, int ksz);
long tvae_conv1_x6_dy_bytes(int B, int C, int R, int n, int ksz, int pad);
int tvae_bank_split3(const float* bank, void* a3, long a3_bytes, int C, int R, int Cin, int ksz, tvae_stream_t stream);
int tvae_conv1_fwd_x6(const float* y, const void* a3, const float* bias, float* out, int B, int Cin, int n, int ksz,
                      int pad, int C, int R, int act, float slope, tvae_stream_t stream);
int tvae_dy_split3(const float* dpre, void* d3, long d3_bytes, int B, int Cin, int n, int ksz, int pad, int C, int R,
                   tvae_stream_t stream);
int tvae_conv1_wgrad_x6(const float* y, const void* d3, float* dbank, float* ws, long ws_floats, int B, int Cin, int n,
                        int ksz, int pad, int C, int R, tvae_stream_t stream);

/* ---- lifting convolution through the frequency domain ("dft") -----------------------------------------------------
 * Same operator as tvae_conv1_fwd / tvae_conv1_wgrad (src/models.py:215) for Cin = 1, evaluated with the circular-
 * correlation theorem on an L x L frame -- ABI 6: L = max(n + pad, ksz) rounded up to a multiple of 4 (tvae_conv1_dft_frame),
 * not the reference's n + 2*pad: the leading and the trailing zero band of the padded image share their storage on the
 * circle, and the result is still exact (csrc/abi_conv_dft.hip: dft_plan) -- DFT of image and filters by direct sums, the
 * spectral contraction as one batched GEMM per call on the split-bf16 dense kernels (10x fewer matrix FLOPs than the direct
 * form at the 64 x 64 shape), the transforms along w on the fp32 matrix pipe.  tvae_conv1_dft_ring: which instance of those
 * transforms the geometry takes (0 = register-staged / generic, else an LDS-DMA ring kernel).  `at` (tvae_conv1_dft_at_floats floats) receives the image
 * spectra in GEMM-operand form in the forward call and is consumed again by the weight gradient of the same step;
 * ws is scratch (tvae_conv1_dft_ws_floats floats); dbias (C floats, may be NULL) receives the bias gradient
 * sum over (image, rotation, position) of dpre, read off the zero-frequency row.  tvae_conv1_dft_supported: 1 if this geometry is handled
 * (single input channel, output side <= 64, frame side <= 126). */
int tvae_conv1_dft_supported(int B, int Cin, int n, int ksz, int pad, int C, int R);
long tvae_conv1_dft_at_floats(int B, int Cin, int n, int ksz, int pad, int C, int R);
long tvae_conv1_dft_ws_floats(int B, int Cin, int n, int ksz, int pad, int C, int R);
int tvae_conv1_dft_frame(int B, int Cin, int n, int ksz, int pad, int C, int R);
int tvae_conv1_dft_ring(int B, int Cin, int n, int ksz, int pad, int C, int R);
int tvae_conv1_fwd_dft(const float* y, const float* bank, const float* bias, float* out, float* at, float* ws,
                       long ws_floats, int B, int Cin, int n, int ksz, int pad, int C, int R, int act, float slope,
                       int parts, tvae_stream_t stream);
int tvae_conv1_wgrad_dft(const float* dpre, const float* at, float* dbank, float* dbias, float* ws, long ws_floats, int B,
                         int Cin, int n, int ksz, int pad, int C, int R, int parts, tvae_stream_t stream);

/* ---- dense layers in the same "x6" arithmetic (nn.Linear of SpatialGenerator, src/models.py:78-93,119-120) -----------
 * tvae_dense_split3: W (row stride ldw) -> cells for A(row, k) = W[row][k] (transpose = 0: forward, rows = out
 *                    features) or A(row, k) = W[k][row] (transpose = 1: data gradient, rows = in features);
 *                    a3 needs tvae_dense_x6_bytes(rows, K) bytes (host query).  scale [K] (optional): A(row, k) is
 *                    multiplied by scale[k] before the split; rowsum [rows] (optional) receives sum_k A(row, k).
 * tvae_linear_fwd_x6 / tvae_linear_dgrad_x6: as tvae_linear_fwd / tvae_linear_dgrad with the split weight instead of
 *                    W (no per-image bias); N must be a multiple of 128 (else hipErrorInvalidValue: use the fp32 entry).
 *                    tvae_linear_fwd_x6 can also apply the NEXT layer when that is the single-output Linear
 *                    (src/models.py:121-123): col_y[n] = col_b[0] + sum_m col_w[m] Y[m][n] (col_w = NULL: off; needs
 *                    M <= 512), which saves a separate pass over Y. */
long tvae_dense_x6_bytes(int rows, int K);
int tvae_dense_split3(const float* W, long ldw, void* a3, long a3_bytes, int rows, int K, int transpose,
                      const float* scale, float* rowsum, tvae_stream_t stream);
/* The same cells in the h3 arithmetic (parts = 2 of the entry points below: TWO fp16 parts per operand, THREE products per
 * block instead of six; every ROW of the operand scaled by its own power of two into fp16's range, undone per output row
 * in the epilogue; accuracy against fp64 at least that of the fp32 matrix pipe, profiles/experiments/f16_split_probe.hip).
 * Same buffer size; one maximum per padded row is kept behind the two part arrays, and the GEMM entry points use the words
 * behind them as scratch (the bound words of a recomputed operand: 4 + K).
 * Entry points / operand forms with an h3 instance: tvae_conv1_fwd_dft / tvae_conv1_wgrad_dft (every geometry),
 * tvae_linear_fwd_x6 with the recomputed first-layer operand (va_xr), tvae_linear_dgrad_x6 in its two-valued form (vg_csum),
 * tvae_linear_wgrad_x6 from sign bits with the recomputed operand.  Elsewhere parts = 2 is rejected (hipErrorInvalidValue)
 * or, inside the *_dft entry points, runs the exact three-part split. */
int tvae_dense_split2h(const float* W, long ldw, void* a3, long a3_bytes, int rows, int K, int transpose,
                       const float* scale, float* rowsum, tvae_stream_t stream);
int tvae_linear_fwd_x6(const void* w3, const float* X, const float* bias, const float* res, float* Y, int M, int N,
                       int K, long ldx, long ldy, int act, float slope, const float* col_w, const float* col_b,
                       float* col_y, const float* va_xr, const float* va_wc, const float* va_bc, const float* va_lb,
                       int va_np, void* sign_bits, int parts, const float* x_amax, float* y_amax, tvae_stream_t stream);
/* y_amax (optional, ABI 7; also the last argument of tvae_linear_dgrad_x6): ONE device word, zeroed by the caller, that
 * receives max |Y| (max |dX|) of what the launch stores, by atomic max from its epilogue -- the measured h3 bound x_amax of the
 * launch that streams this output next, so that every hidden layer of a deep decoder runs the two-part arithmetic, not only
 * the one whose input has an analytic bound.  Needs the output stored and nothing fused behind it (no col_w / sign_bits;
 * data gradient: no in_xr).
 * x_amax (optional, ABI 5): with parts = 2 and X read from memory, ONE device word holding max |X| or an upper bound of it
 * (the h3 scale of the streamed operand); the same argument of tvae_linear_dgrad_x6 (plain form: X = dpre) and a_amax /
 * x_amax of tvae_linear_wgrad_x6 (plain form: max |dpre|, max |X|; two-valued form from sign bits against an operand from
 * memory: x_amax >= max |gy[n] X[k][n]|).  A bound that is 2^j too large costs j of the 16 bits by which an element may lie
 * below its group's maximum before its low part goes subnormal (csrc/conv_x6_kernels.hpp).  ABI 6: x_amax_rows != 0 of
 * tvae_linear_wgrad_x6 -- x_amax then holds K words, one bound per ROW of X (a row of X is a column of dW: a hidden unit far
 * below the others -- dead, or not yet trained -- keeps the full two-part precision relative to ITSELF, as the recomputed form
 * does through dec_l0_bound_kernel's per-unit words). */
int tvae_linear_dgrad_x6(const void* w3t, const float* dpre, const float* add, const float* aux, float* dX, int M,
                         int N, int K, long ldd, long ldx, int mask, float slope, const float* in_xr,
                         const float* in_wc, float* in_gxr, float* in_part, long in_part_floats, const float* vg_wo,
                         const float* vg_gy, const float* vg_csum, const float* in_bc, const float* in_lb, int in_np,
                         float* rs_part, long rs_part_floats, const float* rs_wo, const float* rs_gysum, float* rs_db,
                         float* rs_dwo, int parts, const void* vg_bits, const float* rs_rowdot, const float* rs_bias,
                         const float* x_amax, float* y_amax, tvae_stream_t stream);
/* ABI 5, the two-valued form WITHOUT the saved activation (vg_bits != NULL; dpre may then be NULL): the 0 / 1 operand
 * [H > 0] and the row sums sum_n gy[n] [H[m][n] > 0] come from the sign bits the forward launch stored (tvae_linear_fwd_x6
 * sign_bits; that launch may then be given Y = NULL and never writes H; ABI 6: with Y = NULL AND sign_bits = NULL it is the
 * inference-mode forward -- only the fused column dot col_y leaves the launch), and the weight gradient of the single-output
 * Linear behind the layer, dWo[m] = sum_n gy[n] H[m][n], from the identity act(p) = act'(p) p of LeakyReLU:
 *     rs_dwo[m] = rs_rowdot[m] + rs_bias[m] * sum_n gy[n] act'(H[m][n]),   rs_rowdot[m] = sum_k W[m][k] G[m][k]
 * (G = this layer's weight gradient before its row factor wo[m]: tvae_linear_wgrad_x6 rd_rowdot, which must run first;
 * rs_bias = the layer's bias or NULL).  Requires vg_csum, rs_part .. rs_dwo, N % 32 == 0. */
/* ABI 6: with rs_db = rs_dwo = NULL tvae_linear_dgrad_x6 leaves only the per-tile partial sums in rs_part; the caller totals
 * them with tvae_dgrad_rowsum_total (ntiles = N / 128; rs_rowdot / rs_bias as above, NULL for the form that summed H itself) --
 * on another stream if it likes: the totals are the only part of the data-gradient launch that waits for the weight gradient. */
int tvae_dgrad_rowsum_total(const float* rs_part, int ntiles, int M, const float* rs_wo, const float* rs_gysum, float slope,
                            float* rs_db, float* rs_dwo, const float* rs_rowdot, const float* rs_bias, tvae_stream_t stream);
/* Row sums of the streamed activation (ABI 3; two-valued form only, i.e. vg_csum given; M <= 512): with rs_part
 * [M][N/128][2] (workspace), rs_wo [M] (the single-output Linear's weight, src/models.py:121-123), rs_gysum [1] = sum_n
 * vg_gy[n], the launch also returns rs_db [M] = wo[m] sum_n gy[n] act'(H[m][n]) (bias gradient of the layer that produced
 * H = dpre) and rs_dwo [M] = sum_n gy[n] H[m][n] (weight gradient of that Linear) -- what tvae_dec_out_bwd would otherwise
 * compute in a 2 GB pass of its own over H.  rs_part = NULL: off.
 * tvae_linear_dgrad_x6 can also consume its result for the backward of SpatialGenerator's first layer (no Fourier
 * features, src/models.py:107-118): with in_xr [N][2], in_wc [K][2] it writes the coordinate gradient in_gxr [N][2] and
 * per-128-column panel row sums in_part [N/128][K][3] (K <= 512, panels must not straddle images); dX may then be NULL
 * (never materialised).  tvae_dec_in_total turns the panels into Simg [B][F], dbc [F], dWc [F][2] (cpi panels per image);
 * it CONSUMES part (ABI 4: the first panel of every image is overwritten with the image's sums). */
int tvae_dec_in_total(float* part, int B, int cpi, int F, float* Simg, float* dbc, float* dWc, tvae_stream_t stream);
/* tvae_linear_wgrad_x6: as tvae_linear_wgrad (both operands are split on the fly); needs N % 16 == 0, 16-byte aligned
 * rows and a workspace of tvae_linear_wgrad_x6_ws_floats(M, N, K) floats (hipErrorInvalidValue otherwise: use the fp32
 * entry).  The number of reduction slices -- i.e. the summation order -- depends on the shape only. */
long tvae_linear_wgrad_x6_ws_floats(int M, int N, int K);
int tvae_linear_wgrad_x6(const float* dpre, const float* X, float* dW, float* ws, long ws_floats, int M, int N, int K,
                         long ldd, long ldx, int accumulate, const float* vg_wo, const float* vg_gy, int vg_act,
                         float vg_slope, const float* va_xr, const float* va_wc, const float* va_bc, const float* va_lb,
                         int va_np, const void* vg_bits, int parts, const float* rd_w, long rd_ldw, float* rd_rowdot,
                         const float* a_amax, const float* x_amax, int x_amax_rows, tvae_stream_t stream);
/* rd_rowdot (optional, ABI 5; two-valued LeakyReLU form, accumulate = 0): also returns rd_rowdot[m] = sum_k rd_w[m][k] G[m][k]
 * with G[m][k] = dW[m][k] / wo[m] taken BEFORE the multiplication (exact for wo[m] = 0); rd_w = the layer's weight, [M][K]
 * with row stride rd_ldw. */
/* `parts` (every *_x6 / *_dft compute entry): 3 = the exact three-part bf16 split (six products per block, fp32-equivalent
 * results: the default of the Python layer); 1 = operands rounded to ONE bf16 number (a single product per block, fp32
 * accumulate): the bf16 throughput mode BASELINE.json names for configs 2 and 5 -- about 3 significant digits per
 * product, NOT held to the fp32 parity tolerances.  Anything else: hipErrorInvalidValue.
 * Implicit gradient operand (vg_wo != NULL, both entries): dpre is then NOT the gradient but the saved activation H of
 * the layer in front of the single-output last Linear, and the gradient is formed on the fly,
 * dpre_eff[m][n] = vg_wo[m] * vg_gy[n] * act'(H[m][n])  (act = `mask` for the data gradient, vg_act for the weight
 * gradient), so the [hid][B*n^2] gradient tensor is never written; tvae_dec_out_bwd with D = NULL then only produces
 * the row sums.
 * Two-valued form (LeakyReLU): act' = slope + (1 - slope) [H > 0], so the streamed operand can be the 0 / 1 matrix
 * [H > 0] -- ONE exact bf16 part, three MFMAs per product block instead of six, no split arithmetic -- with the row /
 * column factors moved out of the sum:
 *   data gradient (vg_csum != NULL; w3t = tvae_dense_split3 of W^T with scale = vg_wo, vg_csum = its rowsum; vg_wo unused):
 *       sum_m W[m][k] dpre_eff[m][n] = vg_gy[n] * (slope * vg_csum[k] + (1 - slope) * sum_m (W[m][k] vg_wo[m]) [H[m][n] > 0]);
 *   weight gradient (vg_act = LeakyReLU, chosen by the library):
 *       dW[m][k] = vg_wo[m] * (slope * s[k] + (1 - slope) * sum_n [H[m][n] > 0] vg_gy[n] X[k][n]),  s[k] = sum_n vg_gy[n] X[k][n].
 *   Sign bits: a LeakyReLU tvae_linear_fwd_x6 launch can store [Y > 0] as one bit per element (sign_bits: uint32
 *   [M][N/32], bit n & 31 of word n / 32; N % 32 == 0); passed as vg_bits to tvae_linear_wgrad_x6 the 0 / 1 operand is
 *   read from them -- 1/32 of the bytes, two 4-byte LDS-DMAs per wave and step instead of four 1 KB ones -- and dpre
 *   (the activation) may be NULL.
 * Recomputed first-layer operand (va_xr != NULL in fwd / wgrad, in_bc != NULL in dgrad): the input of the layer is the
 * output of SpatialGenerator's coordinate layer without Fourier features (src/models.py:107-118),
 *   h0[f][n] = act(fma(Wc[f][1], xr[n][1], fma(Wc[f][0], xr[n][0], bc[f])) + LB[n / np][f])
 * (exactly the expression tvae_dec_l0_fwd evaluates), formed inside the kernel from xr [N][2], Wc [K][2], bc [K] and the
 * per-image latent term LB [B][K]; X (resp. aux for the mask) may then be NULL and the [hid][B*n^2] tensor is never
 * stored.  Needs np % 128 == 0 and K <= 512 (fwd / dgrad), np % 4 == 0 (wgrad). */

/* ---- fused skinny ends of the two MLPs: one pass over the 1-2 GB activation instead of 2-3 -------------------------
 * dec_out_bwd: backward of the last decoder layer y = Wo h + bo (SpatialGenerator.forward, src/models.py:121-123),
 *              replaces outer_mask + two rowdot passes:
 *                D[f][n] = (sum_o Wo[o*F+f] gy[n*n_out+o]) * act'(H[f][n]);
 *                tot[0][f] = sum_n D[f][n]  (bias gradient of the layer that produced h);
 *                tot[1+o][f] = sum_n H[f][n] gy[n*n_out+o]  (= dWo[o][f]).          n_out <= 4
 *              part: workspace >= ceil(N/1024)*F*(1+n_out) floats; tot: (1+n_out)*F floats.  D may be NULL (sums only).
 * dec_in_bwd:  backward of the first decoder layer h = act(Wc x' + bc + Wl z) without Fourier features
 *              (src/models.py:107-118), d = pre-activation gradient [F][B*Np]:
 *                gxr[n][j] = sum_f Wc[2f+j] d[f][n];  Simg[b][f] = sum_{n in image b} d[f][n];
 *                dbc[f] = sum_n d[f][n];  dWc[f][j] = sum_n d[f][n] x'[n][j].
 *              part: workspace >= B*ceil(Np/1024)*F*3 floats.
 * heads_fwd:   conv_a / conv_r / conv_z (src/models.py:390-392) as ONE stacked projection with nh = 3+2*z_dim <= 8
 *              rows: Y[j][n] = b[j] + sum_c W[j*C+c] X[c][n].
 * heads_bwd:   its backward, fused with the activation mask of X and the row reductions:
 *                dX[c][n] = act'(X[c][n]) sum_j W[j*C+c] dY[j][n];
 *                tot[j][c] = sum_n dY[j][n] X[c][n] (= dW[j][c], j < nh);  tot[nh][c] = sum_n dX[c][n].
 *              part: workspace >= ceil(N/512)*C*(nh+1) floats; tot: (nh+1)*C floats.
 * All return hipErrorInvalidValue when n_out / nh is out of range or the workspace is too small. */
int tvae_dec_out_bwd(const float* gy, int n_out, const float* Wo, const float* H, long ldh, float* D, long ldd, int F,
                     long N, int act, float slope, float* part, long part_floats, float* tot, tvae_stream_t stream);
int tvae_dec_in_bwd(const float* d, long ldd, const float* xr, const float* Wc, int F, int B, int Np, float* gxr,
                    float* Simg, float* dbc, float* dWc, float* part, long part_floats, tvae_stream_t stream);
int tvae_heads_fwd(const float* W, const float* X, long ldx, const float* bias, float* Y, long ldy, int nh, int C,
                   long N, tvae_stream_t stream);
int tvae_heads_bwd(const float* W, const float* dY, long ldy, const float* X, long ldx, float* dX, long lddx, int nh,
                   int C, long N, int act, float slope, float* part, long part_floats, float* tot,
                   tvae_stream_t stream);

/* ---- attention head: src/models.py:358-401 (prior add, log_softmax, gumbel_softmax, offsets) fused with the
 * pooling / sampling / KL block of eval_minibatch, train_mnist.py:192-231,242-282.
 * heads [3+2*zd][ldh] rows: 0 logit, 1 theta_mu, 2 theta_logstd, 3.. z_mu, 3+zd.. z_logstd (column = img*R*P + j).
 * E [B][R*P] Exp(1) noise, eps_z [B][zd], eps_t [B]; tables p_r [R], off [R], p_tr [R*P], grid [P][2].
 * Outputs: attn, q (log-softmax), a (Gumbel-softmax sample) [B][R*P]; z [B][zd]; theta [B]; dx [B][2]; kl [B]. */
int tvae_attn_head_fwd(const float* heads, long ldh, const float* E, const float* eps_z, const float* eps_t,
                       const float* p_r, const float* off, const float* p_tr, const float* grid, int B, int R, int P,
                       int zd, float sigma_p, float theta_off_scale, float* attn, float* q, float* a, float* z,
                       float* theta, float* dx, float* kl, float* part, long part_floats, tvae_stream_t stream);
/* part (optional workspace, both entries): with few images of very many positions (B < 128, R*P >= 16 384: the galaxy
 * configuration has 8 x 266 256) an image is spread over up to 1024/B workgroups and the partial (max, sum) pairs /
 * pooled sums go through it: forward needs B * G * (7 + 3*(zd+1)) floats, backward 2 * B * G; NULL or too small = one
 * workgroup per image.
 * upstream gz [B][zd], gth [B], gdx [B][2], gkl [B]; optional g_attn/g_q/g_a [B][R*P] (NULL = zero) -> dheads */
int tvae_attn_head_bwd(const float* heads, long ldh, const float* q, const float* a, const float* eps_z,
                       const float* eps_t, const float* p_r, const float* off, const float* p_tr, const float* grid,
                       int B, int R, int P, int zd, float sigma_p, float theta_off_scale, const float* gz,
                       const float* gth, const float* gdx, const float* gkl, const float* g_attn, const float* g_q,
                       const float* g_a, float* dheads, float* part, long part_floats, tvae_stream_t stream);

/* ---- rotation pooling of the translation-attention encoder: fc_r = nn.Linear(R, 1) over the rotation axis of
 * act(conv1(x)), src/models.py:301-304.  A1 [C][B][R][P] (the feature-major conv1 output),
 * X [C][B*P] = fb[0] + sum_r fw[r] A1[.][.][r][.].  Backward: dA1 = fw[r] * dX * act'(A1), dtot [R + 1] = (dfw | dfb);
 * part: workspace >= (R + 1) * min(1024, ceil(C*B*P / 256)) floats.  R <= 16. */
int tvae_rot_pool_fwd(const float* A1, const float* fw, const float* fb, float* X, int C, int B, int R, int P,
                      tvae_stream_t stream);
int tvae_rot_pool_bwd(const float* A1, const float* dX, const float* fw, float* dA1, float* part, long part_floats,
                      float* dtot, int C, int B, int R, int P, int act, float slope, tvae_stream_t stream);

/* ---- encoder tail fused per direction on the split bf16 pipe: conv2 (1x1x1, src/models.py:347-351,356) + the stacked
 * head projection conv_a / conv_r / conv_z (:357-358, 390-392).  C = 128 channels in and out, nh <= 7 head rows,
 * feature-major operands, N = B*R*Ho*Ho columns (any N).  w3 = tvae_dense_split3(W2, rows 128, K 128, transpose 0).
 *   forward: H = act(W2 A1 + b2) [128][N],  heads = Wh H + bh [nh][N]; one pass over A1, one over H.  bits_h / bits_a
 *     (LeakyReLU only, both or neither; [N][4] uint32 each): bit (r & 31) of word r >> 5 of column n = [H[r][n] > 0]
 *     resp. [A1[r][n] > 0] -- all the data gradient needs of the two tensors.  H == NULL (ABI 6): the inference-mode
 *     forward of eval_model / get_latent (train_mnist.py:352-387 under torch.no_grad(), clustering_mnist.py:121-161) --
 *     H is not written at all, only the nh head rows leave the kernel (bits_* NULL as well).
 *   data gradient (LeakyReLU): dA1 = act'(A1) . W2^T (act'(H) . Wh^T dheads) from dheads [nh][N] and the sign words; dH is
 *     never stored.  w3p = tvae_dense_split3 (rows 128, K 128, transpose 0) of W2^T with its columns permuted:
 *     column 16 u + 8 h + j (u < 8, h < 2, j < 8) holds W2[16 u + 8 (j >> 2) + 4 h + (j & 3)][.] (the order in which
 *     the first GEMM's accumulators feed the second; tvae/ops.py:_enc_tail_perm); wh3 = tvae_dense_split3(Wh, ldw 128,
 *     rows 128, K nh, transpose 1). */
int tvae_enc_tail_fwd_x6(const void* w3, const float* A1, long lda, const float* b2, const float* Wh, const float* bh,
                         int nh, float* H, long ldh, float* heads, long ldo, void* bits_h, void* bits_a, int C, long N,
                         int act, float slope, int parts, const float* amax_a1, tvae_stream_t stream);
/* parts = 2 (h3): w3 = tvae_dense_split2h cells and amax_a1 = C device words holding max |A1[c][:]| per channel (or upper
 * bounds; ABI 5) -- tvae_conv1_fwd_dft leaves them in the LAST C words of the buffer sized by tvae_conv1_dft_at_floats.  The
 * forward scales A1 by the largest of them (the channels are its reduction index), tvae_enc_tail_wgrad_x6 every channel by
 * its own (a channel is a column of dW2).  tvae_enc_tail_dgrad_x6 with parts = 2: w3p = tvae_dense_split2h cells, wh3 stays
 * tvae_dense_split3 (the skinny GEMM runs the exact split); its streamed operand is scaled per 32-column chunk inside the
 * kernel, no maximum is passed. */
int tvae_enc_tail_dgrad_x6(const void* w3p, const void* wh3, const float* dheads, long ldd, int nh, const void* bits_h,
                           const void* bits_a, float* dA1, long lda, int C, long N, float slope, int parts,
                           tvae_stream_t stream);
/* tvae_enc_tail_wgrad_x6 (ABI 3): weight gradient of conv2 in ONE pass, dH never stored (autograd of nn.Conv3d(C, C, 1)
 * behind the 1x1x1 heads, src/models.py:347-358,390-392):  dW2[c2][c] = sum_n dH[c2][n] A1[c][n] with
 * dH[c2][n] = act'(H[c2][n]) sum_h Wh[h][c2] dheads[h][n] formed on the fly from the head gradients and the sign words
 * of H (bits_h, as stored by tvae_enc_tail_fwd_x6; LeakyReLU).  C = 128, nh <= 7, N % 32 == 0, 16-byte aligned rows;
 * ws: tvae_enc_tail_wgrad_x6_ws_floats(N) floats of per-workgroup slabs, added in a fixed order.  (dWh and db2 come from
 * tvae_heads_bwd with dX = NULL.) */
long tvae_enc_tail_wgrad_x6_ws_floats(long N);
int tvae_enc_tail_wgrad_x6(const float* A1, long lda, const float* dheads, long ldd, int nh, const void* bits_h,
                           const float* Wh, float* dW2, float* ws, long ws_floats, int C, long N, float slope, int parts, const float* amax_a1,
                           tvae_stream_t stream);
/* ---- encoder tail with MANY head rows (ABI 7; 8 <= nh <= 128, e.g. the galaxy configuration's z_dim = 50 -> 103 rows:
 * reference train_galaxy.py:412-420, src/models.py:347-358,390-392): both directions as two chained 128 x 128 split-pipe
 * GEMMs per 32-column chunk with a register hand-off (csrc/enc_tail_wide_kernels.hpp); parts = 2 (h3) or 1 (bf16) -- both
 * weights stay in LDS, which the exact three-part split does not fit.  C = 128.
 *   forward: w3 = cells of W2 (tvae_dense_split2h / split3 with one part used, rows = K = 128); wh3 = cells of Wh with its K
 *     columns in the order slot 16 u + 8 h + j <- row 16 u + 8 (j >> 2) + 4 h + (j & 3) (rows = nh, K = 128); H may be NULL
 *     (inference); bits_h / bits_a (LeakyReLU, both or neither): the sign words tvae_enc_tail_fwd_x6 writes; amax_a1 as there.
 *   data gradient (LeakyReLU): wht3 = cells of Wh^T (transpose = 1: rows = 128, K = nh); w3p = cells of W2^T with the same
 *     K order (as tvae_enc_tail_dgrad_x6); dH (optional) receives act'(H) . Wh^T dheads for the weight gradients
 *     (dW2 = dH A1^T, db2 = rowsum dH); amax_dheads (parts = 2): ONE word >= max |dheads| (tvae_rowdot_seg amax). */
int tvae_enc_tail_wide_max_rows(void);
int tvae_enc_tail_fwd_wide(const void* w3, const void* wh3, const float* A1, long lda, const float* b2, const float* bh, int nh,
                           float* H, long ldh, float* heads, long ldo, void* bits_h, void* bits_a, int C, long N, int act,
                           float slope, int parts, const float* amax_a1, tvae_stream_t stream);
int tvae_enc_tail_dgrad_wide(const void* wht3, const void* w3p, const float* dheads, long ldd, int nh, const void* bits_h,
                             const void* bits_a, float* dH, long ldh, float* dA1, long lda, int C, long N, float slope,
                             int parts, const float* amax_dheads, tvae_stream_t stream);
/* tvae_enc_tail_wgrad_wide (ABI 7): the two weight gradients of that tail, dW2 = dH A1^T and dWh = dheads H^T (autograd of the
 * 1x1x1 convolutions, src/models.py:347-358,390-392), as ONE cooperative reduction over the columns each:
 * dW[r][c] = sum_n D[r][n] A[c][n], D [rows_d <= 128][N], A [128][N], both stored tensors streamed by LDS-DMA, h3 arithmetic
 * (amax_d: one word >= max |D|; amax_a: 128 floats, one bound per row of A).  dW is [128][128] (rows >= rows_d unspecified);
 * ws as tvae_enc_tail_wgrad_x6_ws_floats(N); N % 32 == 0, C == 128. */
int tvae_enc_tail_wgrad_wide(const float* D, long ldd, int rows_d, const float* A, long lda, float* dW, float* ws,
                             long ws_floats, int C, long N, const float* amax_d, const float* amax_a, tvae_stream_t stream);
/* ---- inference epilogue: get_latent, clustering_mnist.py:123-161 (argmax over (r,h,w) of attn, gather of
 * (z_mu, exp(z_logstd)) and theta_mu there, softmax-expected translation).  zc [B][2*zd], theta_mu [B], dx [B][2]. */
int tvae_get_latent(const float* heads, long ldh, const float* p_r, const float* off, const float* grid, int B, int R,
                    int P, int zd, float theta_off_scale, float* zc, float* theta_mu, float* dx, tvae_stream_t stream);

/* ---- coordinate transform: train_mnist.py:222,234-239.  xc [Np][2], dx [B][2], theta [B] -> xr [B][Np][2] ---- */
int tvae_coord_fwd(const float* xc, const float* dx, const float* theta, float* xr, int B, int Np,
                   tvae_stream_t stream);
int tvae_coord_bwd(const float* xc, const float* dx, const float* theta, const float* gxr, float* gdx, float* gtheta,
                   int B, int Np, tvae_stream_t stream);

/* ---- spatial decoder ends: src/models.py:95-123 ----
 * dec_l0_fwd:  h[f][pix] = act(Wc[f][0] x0 + Wc[f][1] x1 + bc[f] + LB[img][f])      (coord_linear, in_dim = 2)
 * latent_bias: LB[img][f] = sum_d Wl[f][d] z[img][d]                                 (latent_linear, models.py:111-116)
 * latent_bwd:  from S[img][f] = sum_pix dpre0[f][pix]: dWl[f][d], dz[img][d]
 * fourier_fwd: feat[f][pix] = cos((Wf[f]/sigma) . x + bf[f])                         (models.py:53-58)
 * fourier_bwd: gxr[pix][j] = sum_f -sin(arg) (Wf[f][j]/sigma) dfeat[f][pix] */
int tvae_dec_l0_fwd(const float* xr, const float* Wc, const float* bc, const float* LB, float* h, long ldh, int F,
                    long Ntot, int Np, int act, float slope, tvae_stream_t stream);
int tvae_latent_bias(const float* Wl, const float* z, float* LB, int B, int F, int zd, tvae_stream_t stream);
int tvae_latent_bwd(const float* S, const float* Wl, const float* z, float* dWl, float* dz, int B, int F, int zd,
                    tvae_stream_t stream);
int tvae_fourier_fwd(const float* xr, const float* Wf, const float* bf, float sigma, float* feat, long ld, int F,
                     long Ntot, tvae_stream_t stream);
int tvae_fourier_bwd(const float* xr, const float* Wf, const float* bf, float sigma, const float* dfeat, long ld,
                     int F, long Ntot, float* gxr, tvae_stream_t stream);

/* ---- likelihoods over flat per-image vectors: kind 0 BCE-with-logits (train_mnist.py:288-291,
 * train_galaxy.py:288-292), 1 Gaussian (train_particles.py:338), 2 Gaussian with learned log-variance
 * (train_particles.py:293-296,336; mu = yh[i], logvar = yh[L+i]).  lp [B] = per-image log-likelihood. */
int tvae_loglik_fwd(const float* yh, const float* y, float* lp, int B, int L, int kind, tvae_stream_t stream);
int tvae_loglik_bwd(const float* yh, const float* y, const float* glp, float* gyh, int B, int L, int kind,
                    tvae_stream_t stream);

/* ELBO scalars of a minibatch (train_mnist.py:282,291-292) in one launch: logp[0] = float32 mean of lp [B], kld[0] = float64
 * mean of kl [B], elbo[0] = logp - kld (float64); and the chain rule of all three in one more (NULL upstream gradient = 0):
 * g_lp[b] = (g_elbo + g_logp) / B, g_kl[b] = (g_kld - g_elbo) / B.  (ABI 6) */
int tvae_elbo_reduce(const float* lp, const float* kl, int B, double* elbo, float* logp, double* kld, tvae_stream_t stream);
int tvae_elbo_reduce_bwd(const double* g_elbo, const float* g_logp, const double* g_kld, int B, float* g_lp, float* g_kl,
                         tvae_stream_t stream);

/* ---- gradient all-reduce of the data-parallel step (the reference is single-GPU; SURVEY 8b / 8e: one sum all-reduce of the
 * flat fp32 gradient buffer per step over RCCL).  RCCL is resolved at run time (dlopen; inside a PyTorch process the instance
 * torch already loaded), so libtvae_hip.so has no link dependency on it: tvae_rccl_available() == 0 when none is found and the
 * other entry points then return 100001.  The communicator belongs to the caller: one rank fills a 128-byte id
 * (tvae_rccl_unique_id), distributes it by any means (tvae/dp.py: torch.distributed broadcast), every rank calls
 * tvae_rccl_comm_init on its current device (a collective), tvae_allreduce_flat sums `count` floats in place on `stream`
 * (asynchronous; the two buckets of tvae/optim.py are two calls on segments of the buffer), tvae_rccl_comm_destroy frees it.
 * The default data-parallel path of this repo calls RCCL through torch.distributed ("nccl" backend); TVAE_DP_ABI=1 routes
 * the two buckets through these entry points instead (tvae/dp.py: GradReducer).  (ABI 6) */
int tvae_rccl_available(void);
int tvae_rccl_unique_id(void* id128);
int tvae_rccl_comm_init(void** comm, int nranks, const void* id128, int rank);
int tvae_allreduce_flat(void* comm, float* buf, long count, tvae_stream_t stream);
int tvae_rccl_comm_destroy(void* comm);

/* ---- particle likelihood tail: train_particles.py:298-338 ----
 * ctf_corr: per-image depthwise cross-correlation out[b] = in[b] (*) ctf[b] with an odd kc x kc filter and zero padding
 * kc/2 (F.conv2d(y_mu.view(1,B,n,n), ctf, padding=pad, groups=B), :298-302); flip = 1 uses the 180-degree rotated
 * filter, i.e. the gradient w.r.t. `in`.  in/out [B][n*n], ctf [B][kc*kc].
 * loglik_masked: Gaussian log-likelihood restricted to a circle of `radius` pixels centred at the inferred
 * translation dx [B][2] / spacing (:309-333,338); masked pixels get no gradient. */
int tvae_ctf_corr(const float* in, const float* ctf, float* out, int B, int n, int kc, int flip, tvae_stream_t stream);
int tvae_loglik_masked_fwd(const float* yh, const float* y, const float* dx, float inv_spacing, float radius, int B,
                           int n, float* lp, tvae_stream_t stream);
int tvae_loglik_masked_bwd(const float* yh, const float* y, const float* dx, float inv_spacing, float radius, int B,
                           int n, const float* glp, float* gyh, tvae_stream_t stream);

/* ---- fused Adam over the flat parameter buffer: torch.optim.Adam defaults, train_mnist.py:579,323 ----
 * bc1 = 1 - b1^t, bc2_sqrt = sqrt(1 - b2^t) computed by the host; grad_scale folds the DP average. */
int tvae_adam_flat(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                   float bc1, float bc2_sqrt, float grad_scale, tvae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TVAE_HIP_H */
