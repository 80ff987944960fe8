/* tamtr_hip.h - C ABI of libtamtr_hip.so: hand-written CDNA4 (gfx950 / MI355X) kernels for the TAM-TR
 * text-image attention hot path (BTA-PAN max-sigmoid text gate + MEH deformable-attention decoder).
 *
 * The reference (Xjh-UCAS/TAM-TR) is 100 % Python and has no FFI of its own; each entry point
 * below therefore cites the reference *Python* interface whose arithmetic it replaces (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM), row-major/contiguous in the layout stated per function;
 *   - `dtype`: element type of the activation tensors marked (T): TAMTR_F32 or TAMTR_BF16.  Index/offset/weight
 *     side inputs and all accumulation are always fp32;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Functions only enqueue work: no
 *     allocation, no synchronisation, no mutable globals (three A/B switches, TAMTR_GEMM, TAMTR_SELFATTN_SCALAR and TAMTR_SCAN_FWD4, are read from the
 *     environment ONCE, into constants; the LDS-size attribute of the kernels that need > 64 KB is set on every call, i.e. for whichever
 *     device is current) - they are safe to call from several host threads on different streams and devices and can be captured into a
 *     hipGraph;
 *   - return value: 0 = enqueued; TAMTR_EINVAL (bad argument), TAMTR_EUNSUP (shape/dtype outside what the kernels
 *     are built for), TAMTR_ELAUNCH (HIP reported a launch error).  Nothing is ever thrown across the ABI.
 */
#ifndef TAMTR_HIP_H
#define TAMTR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAMTR_F32 0
#define TAMTR_BF16 1

#define TAMTR_OK 0
#define TAMTR_EINVAL (-1)
#define TAMTR_EUNSUP (-2)
#define TAMTR_ELAUNCH (-3)

/* ABI version, bumped on any signature change. */
int tamtr_abi_version(void);

/* ---------------------------------------------------------------------------------------------------------------
 * a-1  Max-sigmoid text gate.   Replaces the body of MaxSigmoidAttnBlock.forward,
 *      ultralytics/nn/extra_modules/block.py:217-226 (dup ultralytics/nn/modules/block.py:672-681):
 *          aw  = sigmoid( max_n( <x[b,m,:,p], gk[b,n,m,:]> ) / sqrt(hc) + bias[m] ) * scale
 *          out = v * aw           (broadcast over the hc channels of head m)
 *      x   (T) [B, nh*hc, HW]   NCHW feature map (embed == x: `ec` is None in every TAM-TR instance)
 *      gk  f32 [B, T, nh*hc]    guide after the `gl` Linear (block.py:212-213)
 *      bias f32 [nh]            scale: python float (1.0 in TAM-TR)
 *      v   (T) [B, nh*hc, HW]   proj_conv(x) (Conv3x3+BN, block.py:223)
 *      out (T) [B, nh*hc, HW]
 *      aw  f32 [B, nh, HW]      saved gate (post-sigmoid, pre-scale)      } consumed by _bwd
 *      arg i32 [B, nh, HW]      argmax text index per pixel and head      }
 */
int tamtr_maxsigmoid_gate_fwd(const void* x, const float* gk, const float* bias, const void* v, void* out, float* aw,
                              int32_t* arg, int B, int nh, int hc, int HW, int T, float scale, int dtype, void* stream);

/*      Backward of the above.  dout (T) [B,C,HW] ->
 *      dx (T) [B,C,HW]    gradient through the embed operand (add to the conv's input gradient on the host side)
 *      dv (T) [B,C,HW]    gradient of proj_conv's output
 *      dlogit f32 [B,nh,HW]  d loss / d (max_n <x,gk>)  (host reduces it into dgk / dbias with one small batched GEMM)
 */
int tamtr_maxsigmoid_gate_bwd(const void* dout, const void* x, const float* gk, const void* v, const float* aw,
                              const int32_t* arg, void* dx, void* dv, float* dlogit, int B, int nh, int hc, int HW, int T,
                              float scale, int dtype, void* stream);
/*      Channels-last forward with the branches' BatchNorm folded into the load (`self.proj_conv` = Conv3x3 + BN,
 *      block.py:205,223; `self.ec`, block.py:203): e (T) [B*HW, C] with row pitch ld_e elements (embed; a channel slice of a wider
 *      NHWC map is fine), v (T) [B*HW, C] packed = the RAW convolution output of the value branch; mean_rstd_* f32 [C][2] (batch mean,
 *      1/sqrt(var + eps), as tamtr_bncl_stats writes them), gamma_* / beta_* f32 [C]; a NULL mean_rstd_* means "already normalised /
 *      no BatchNorm" for that operand.  out (T) [B*HW, C] packed; aw / arg as above or NULL when no backward will follow.
 *      C = nh*hc <= 512, C/8 and hc/8 powers of two; e, v, out, gk 16-byte aligned, ld_e % 8 == 0 (bf16) / % 4 == 0 (f32): a lane
 *      moves the 8 channels it owns as 16-byte pieces, four pixel rows of them in flight (round 4).
 */
int tamtr_maxsigmoid_gate_cl_fwd(const void* e, long long ld_e, const float* mean_rstd_e, const float* gamma_e, const float* beta_e,
                                 const void* v, const float* mean_rstd_v, const float* gamma_v, const float* beta_v, const float* gk,
                                 const float* bias, void* out, float* aw, int32_t* arg, int B, int nh, int hc, int HW, int T, float scale,
                                 int dtype, void* stream);

/*      next-3: the value branch itself.  Replaces `self.proj_conv(x)` = Conv(c1, c2, k=3, s=1, act=False) = nn.Conv2d(3x3, stride 1,
 *      pad 1, no bias) + nn.BatchNorm2d in training mode (ultralytics/nn/extra_modules/block.py:205,223; Conv: nn/modules/conv.py)
 *      for channels-last bf16 maps: an implicit-GEMM MFMA convolution whose epilogue leaves the BatchNorm's batch statistics, to be
 *      followed by tamtr_maxsigmoid_gate_cl_fwd, which applies the affine in its load.  Forward only (TIAGELAN discards the gate: SURVEY D2).
 *      x   bf16 [B*H*W, C1] with row pitch ld_x elements (a channel slice of a wider NHWC map is fine; 16-byte aligned, ld_x % 8 == 0)
 *      wpk bf16 [C1/32][9][C2][32]   the weight [C2][C1][3][3] repacked by tamtr_conv3x3_pack_weight (w: TAMTR_F32 or TAMTR_BF16)
 *      y   bf16 [B*H*W, C2] packed   the RAW convolution output (fp32 accumulation, rounded once)
 *      partials f32 [C2][S][3], S = tamtr_conv3x3_tiles(B, H, W): (count, mean, M2) of the stored values per pixel tile
 *      mean_rstd f32 [C2][2] (batch mean, 1/sqrt(biased var + eps)); running_mean / running_var f32 [C2] or both NULL: updated with
 *      `momentum` (unbiased variance), as nn.BatchNorm2d does.  mean_rstd NULL: convolution + partials only.
 *      C1 % 32 == 0, C2 % 64 == 0.  tamtr_bn_finalize: the per-channel combine on its own (partials [C][S][3] -> mean_rstd, running update).
 */
int tamtr_conv3x3_tiles(int B, int H, int W);
int tamtr_conv3x3_pack_weight(const void* w, void* wpk, int C1, int C2, int dtype, void* stream);
int tamtr_conv3x3_cl_stats_fwd(const void* x, long long ld_x, const void* wpk, void* y, float* running_mean, float* running_var,
                               float* mean_rstd, float* partials, int B, int H, int W, int C1, int C2, float eps, float momentum,
                               void* stream);
int tamtr_bn_finalize(const float* partials, float* mean_rstd, float* running_mean, float* running_var, int C, int S, float eps,
                      float momentum, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * a-6  Multi-scale deformable attention core.  Replaces multi_scale_deformable_attn_pytorch,
 *      ultralytics/nn/modules/utils.py:42-89 (3x F.grid_sample(bilinear, zeros, align_corners=False) + weighted sum).
 *      value (T) [B, L, M, D]           L = sum_l H_l*W_l, levels concatenated fine -> coarse
 *      shapes i32 [nl, 2]  (HOST pointer) (H_l, W_l)
 *      loc   f32 [B, Q, M, nl, P, 2]    sampling locations (x, y) in [0,1] image coordinates
 *      aw    f32 [B, Q, M, nl, P]       attention weights (already softmaxed over nl*P)
 *      out   (T) [B, Q, M*D]
 *      D must be a multiple of 4 (f32) / 8 (bf16) and <= 256; nl <= 8.
 */
int tamtr_msdeform_attn_fwd(const void* value, const int32_t* shapes_host, const float* loc, const float* aw, void* out,
                            int B, int L, int M, int D, int Q, int nl, int P, int dtype, void* stream);

/*      Backward: gout (T) [B,Q,M*D] -> gvalue f32 [B,L,M,D] (ACCUMULATED with float atomics: caller zeroes it),
 *      gloc f32 [B,Q,M,nl,P,2], gaw f32 [B,Q,M,nl,P].
 */
int tamtr_msdeform_attn_bwd(const void* gout, const void* value, const int32_t* shapes_host, const float* loc,
                            const float* aw, float* gvalue, float* gloc, float* gaw, int B, int L, int M, int D, int Q,
                            int nl, int P, int dtype, void* stream);

/*      Backward without atomics: the same gloc / gaw, and gvalue (T, the VALUE's dtype) written exactly once per element as an
 *      ordered sum - the caller neither zeroes it nor casts it, and two calls on the same inputs give the same bits (the
 *      reference trains with deterministic=True: ultralytics/cfg/default.yaml:26, utils/torch_utils.py:371-389; torch's own
 *      grid_sample backward is the atomic scatter this replaces).  Per (image, head, level) the Q*P*4 bilinear corners are sorted
 *      by destination row in LDS and every row sums its run.  gvalue rows have pitch ldg elements (M*D for a plain [B,L,M,D]
 *      tensor).  Limits: Q*P*4 <= 8192, D % 8 == 0, D <= 256, H*W < 2^19 - 1 per level; TAMTR_EUNSUP otherwise.
 *      colw f32 [B, Q, M] or NULL (ABI 34): per (image, query, head) the total weight it put ON the map - the sum over levels, points
 *      and the corners that lie on the map of aw * bilinear weight (1 when no corner falls off).  The column sums of gvalue over
 *      its B*L rows are then sum_{b,q} gout[b,q,m,:] * colw[b,q,m]: the bias gradient of the linear layer that produced `value`
 *      (MSDeformAttn.value_proj, nn/modules/transformer.py:273-275) from Q-sized operands, without a pass over the 550 MB gvalue.
 */
int tamtr_msdeform_attn_bwd_sorted(const void* gout, const void* value, const int32_t* shapes_host, const float* loc,
                                   const float* aw, void* gvalue, float* gloc, float* gaw, float* colw, int B, int L, int M,
                                   int D, int Q, int nl, int P, long long ldg, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * a-8  Text-contrastive logits.  Replaces ContrastiveHeadMLP.forward, ultralytics/nn/modules/block.py:534-541:
 *          logits[b,q,k] = < x[b,q,:]/max(|x|,1e-12), w[b,k,:]/max(|w|,1e-12) > * exp(logit_scale) + bias
 *      x (T) [B,Q,C]; w f32 [B,K,C]; logit_scale, bias: f32 device scalars; logits f32 [B,Q,K];
 *      xinv f32 [B,Q], winv f32 [B,K]: saved reciprocal norms (for _bwd).   C % 64 == 0, C <= 2048, K <= 128.
 */
int tamtr_contrastive_logits_fwd(const void* x, const float* w, const float* logit_scale, const float* bias, float* logits,
                                 float* xinv, float* winv, int B, int Q, int K, int C, int dtype, void* stream);

/*      Backward: g f32 [B,Q,K] -> dx (T) [B,Q,C], dwhat f32 [B, S, K, C] with S = tamtr_contrastive_bwd_slabs(Q): every
 *      workgroup STORES the partial sum of its query rows into its own slab (no atomics between workgroups; for K <= 16 none
 *      inside either, so the result is bitwise reproducible); the caller adds the S slabs.  The sum is the gradient w.r.t. the
 *      *normalised* text rows, which the host projects through the normalisation - a [B,K,C] elementwise op.
 */
int tamtr_contrastive_bwd_slabs(int Q);
int tamtr_contrastive_logits_bwd(const float* g, const void* x, const float* w, const float* logit_scale, const float* xinv,
                                 const float* winv, void* dx, float* dwhat, int B, int Q, int K, int C, int dtype,
                                 void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * a-5  MEH cross-attention value projection (the dominant dense contraction, M = B*L = 537 600 at 640^2 bs 16).
 *      Replaces `value = self.value_proj(value)`, ultralytics/nn/modules/transformer.py:273 (nn.Linear(512,512)),
 *      and any other y = x @ W^T + b of the head with N, K multiples of 128/64:
 *          Y[M,N] = X[M,K] @ W[N,K]^T + bias[N]        bf16 in, fp32 MFMA accumulate, bf16 out
 *      X bf16 [M,K] row-major, W bf16 [N,K] row-major (nn.Linear layout), bias f32 [N] or NULL, Y bf16 [M,N].
 *      Requires K % 64 == 0, N % 128 == 0; any M >= 1.
 */
int tamtr_linear_bf16(const void* X, const void* W, const float* bias, void* Y, int M, int N, int K, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * a-7  Small-Q masked multi-head self-attention core.  Replaces the attention inside nn.MultiheadAttention as called
 *      at ultralytics/nn/modules/transformer.py:546 (after the packed in-projection, before out_proj):
 *          O[b,:,h,:] = softmax( Qh Kh^T / sqrt(dh) + mask ) Vh
 *      q,k,v (T): row (b, i) of head h starts at ptr + (b*Q + i)*ld + h*dh  (ldq/ldk/ldv in elements, multiples of 4:
 *      views into the packed in-projection output need no copy); mask_bits u32 [Q, ceil(Q/32)] or NULL, bit j%32 of word
 *      j/32 of row i set = query i may NOT attend to key j; o (T) [B,Q,nh*dh] contiguous; lse f32 [B,nh,Q] saved
 *      log-sum-exp.  dh in {32, 64}, Q <= 4096.
 */
int tamtr_selfattn_fwd(const void* q, const void* k, const void* v, const uint32_t* mask_bits, void* o, float* lse, int B,
                       int Q, int nh, int dh, int ldq, int ldk, int ldv, int dtype, void* stream);
/*      Backward: go (T) [B,Q,nh*dh] -> gq, gk, gv (T) [B,Q,nh*dh] contiguous; delta_ws f32 [B,nh,Q] caller workspace. */
int tamtr_selfattn_bwd(const void* go, const void* q, const void* k, const void* v, const void* o, const float* lse,
                       const uint32_t* mask_bits, void* gq, void* gk, void* gv, float* delta_ws, int B, int Q, int nh, int dh,
                       int ldq, int ldk, int ldv, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * a-9  Selective scan (S6), replacing the external CUDA extension selective_scan_cuda_core.fwd/bwd that the reference
 *      calls at ultralytics/nn/extra_modules/VManba/csms6s.py:257,267 (contract: vmamba.py:962-990), fp32:
 *          dt = softplus(delta + dbias);  h_t = exp(dt_t A) h_{t-1} + dt_t B_t u_t;  y_t = <C_t, h_t> + D u_t
 *      u, delta f32 [B, KD, L]; A f32 [KD, N]; Bm, Cm f32 [B, K, N, L]; D, dbias f32 [KD]; y f32 [B, KD, L].
 *      N == 16.  hstate f32 [B, KD, nchunk, N]: state checkpoints saved for _bwd (the state after every `chunk` steps),
 *      nchunk = ceil(L / chunk) with `chunk` returned by tamtr_selective_scan_chunk() (64 since ABI 23).
 *      xmode = 0: plain contract above.  xmode = 1 ("cross-scan layout", K must be 4; replaces the 4x materialisation of
 *      CrossScan, csms6s.py:4-14): u is [B, 2, Dk, L] = (row-major, column-major) flattenings of the map, direction k reads
 *      u[:, k & 1]; directions k >= 2 are the reversed scans and walk EVERY time-indexed buffer (u, delta, B, C, y and the
 *      gradients) back to front, i.e. those buffers are stored in the un-reversed order of direction k - 2.
 */
int tamtr_selective_scan_chunk(void);
int tamtr_selective_scan_fwd(const float* u, const float* delta, const float* A, const float* Bm, const float* Cm,
                             const float* D, const float* dbias, float* y, float* hstate, int B, int K, int Dk, int N, int L,
                             int xmode, void* stream);
/*      Backward: gy f32 [B,KD,L] -> gu, gdelta f32 [B,KD,L] (in xmode gu is per direction, stored un-reversed: the host adds
 *      gu[:, k] + gu[:, k+2] to get the gradient of u[:, k]); gB, gC f32 [B,K,N,L] (plain stores);
 *      grow f32 [B, KD, S], S = tamtr_selective_scan_row_sums() (= 64): per image and row the sums over time
 *          grow[b, kd, 0:16] = dA[kd, :] | [16:16+R] = d(Wdt)[kd, :] (dtproj variant) | [48] = dD[kd] | [49] = d(dbias)[kd]
 *      written with plain stores (one writer per (b, kd)); the caller adds the B images - in a fixed order, so the result is bitwise
 *      reproducible (ABI <= 21 accumulated gA / gD / gdbias / gWdt over the batch with float atomics).  ws: caller workspace of
 *      2 * tamtr_selective_scan_bwd_slabs(Dk) * B*K*N*L floats (per-workgroup partial dB/dC slabs, summed in slab order by a second
 *      kernel; the dtproj variant reuses it for the row-split partials of gdtr on short sequences).
 *      xmode = 3 (backward only): as xmode = 1 and gy is ALSO in pair layout [B, 2, Dk, L] - the gradient of the
 *      cross-merged map (CrossMerge, csms6s.py:26-34) in its row-major and column-major flattening; direction k reads
 *      gy[:, k & 1], so the four per-direction gradient planes are never materialised.
 */
int tamtr_selective_scan_bwd_slabs(int Dk);
int tamtr_selective_scan_row_sums(void);
int tamtr_selective_scan_bwd(const float* gy, const float* u, const float* delta, const float* A, const float* Bm,
                             const float* Cm, const float* D, const float* dbias, const float* hstate, float* gu,
                             float* gdelta, float* grow, float* gB, float* gC, float* ws, int B, int K, int Dk, int N, int L,
                             int xmode, void* stream);

/*      Fused dt projection (replaces the `dts = einsum("bkrl,kdr->bkdl")` of SS2D.forward_corev2, vmamba.py:972, whose
 *      [B, 4*d_inner, L] result is never materialised): delta[b, k*Dk+d, t] = sum_r Wdt[k*Dk+d, r] * dtr[b, k, r, t] is formed
 *      inside the scan.  dtr f32 [B, K, R, L], Wdt f32 [KD, R], R <= 32.  Backward additionally returns gdtr f32 [B,K,R,L]
 *      (plain stores) and d(Wdt) inside grow (see above); gdelta_ws: caller workspace [B,KD,L] - f32, or bf16 with ws_bf16 = 1 (L % 4 == 0):
 *      d(delta) is only the operand of gdtr = Wdt^T d(delta) there, so in bf16 mode - where the caller rounds gdtr to bf16 anyway - the
 *      workspace, the largest buffer the backward writes and re-reads, can be half the size.
 *      bf16 PLANES (bf16 mode, L % 4 == 0): with planes_bf16 = 1 (forward) / bit 1 of bf16_flags (backward; bit 0 = the d(delta) workspace
 *      above, required with it) the big time-indexed operands that cross HBM - u and y in the forward; gy, u and gu in the backward - are
 *      bf16 arrays of the same shapes.  The recurrence, the states, the checkpoints, dtr / B / C and every gradient sum stay f32: only the
 *      1.7 GB planes of a level are narrowed (what torch autocast makes of `u` anyway: the depthwise convolution's output is bf16 there).
 */
int tamtr_selective_scan_dtproj_fwd(const void* u, const float* dtr, const float* Wdt, const float* A, const float* Bm,
                                    const float* Cm, const float* D, const float* dbias, void* y, float* hstate, int B, int K,
                                    int Dk, int N, int R, int L, int xmode, int planes_bf16, void* stream);
int tamtr_selective_scan_dtproj_bwd(const void* gy, const void* u, const float* dtr, const float* Wdt, const float* A,
                                    const float* Bm, const float* Cm, const float* D, const float* dbias, const float* hstate,
                                    void* gu, void* gdelta_ws, float* gdtr, float* grow, float* gB, float* gC, float* ws, int B,
                                    int K, int Dk, int N, int R, int L, int xmode, int bf16_flags, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * a-10 / next-4  Device-side Hungarian assignment.  Replaces `C.cpu()` + scipy.optimize.linear_sum_assignment per
 *      image in HungarianMatcher.forward, ultralytics/models/utils/ops.py:98-119 (and its four host syncs per step).
 *      Same solver as scipy's (shortest augmenting paths, float64 duals, same scan order and tie rule): the pairs
 *      returned are scipy's, in scipy's order (ascending query index inside each image).
 *      cost  f32 [bs, nq, G]         finite matching costs, G = sum(group_sizes); image b owns columns
 *                                    [off_b, off_b + group_sizes[b])
 *      group_sizes_host i32 [bs]     HOST array (boxes per image; bs <= 256)
 *      batch_idx/query_idx/gt_idx i64 [sum_b min(nq, group_sizes[b])]   outputs: image, query and GLOBAL box index of
 *                                    every matched pair (query_idx = gt_idx = -1 for an image whose costs are not finite)
 *      TAMTR_EUNSUP when one image needs more than 64 KB of LDS (about nq + boxes > 3000).
 */
int tamtr_lsap_assign(const float* cost, const int32_t* group_sizes_host, int bs, int nq, int G, int64_t* batch_idx,
                      int64_t* query_idx, int64_t* gt_idx, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * next-3  CPAM gates.  Replaces CPAM.forward after its max-pool, ultralytics/nn/extra_modules/block.py:271-308:
 *          c   = sigmoid(interpolate(p, scale_factor=2, mode='bilinear', align_corners=False)) * x,  p = MaxPool2d(3,2,1)(x)
 *          out = cat_g( sigmoid(max over the channels of chunk g of c) * c_g ),  g = 0..7  (c.chunk(8, 1))
 *      x, out (T) [B, C, H, W]     H, W even (odd maps make the reference fail too), C % 8 == 0
 *      p      (T) [B, C, H/2, W/2] pooled map (the caller runs the max-pool and keeps its indices for the backward)
 *      s2    f32 [B, 8, H, W]      saved sigmoid of the chunk max     } consumed by _bwd
 *      arg   i32 [B, 8, H, W]      channel (inside the chunk) holding the max }
 */
int tamtr_cpam_fwd(const void* x, const void* p, void* out, float* s2, int32_t* arg, int B, int C, int H, int W, int dtype,
                   void* stream);

/*      Backward.  gout (T) [B,C,H,W] ->
 *      dx_direct (T) [B,C,H,W]    dL/dx through the product sigmoid(up(p)) * x
 *      dp        (T) [B,C,H/2,W/2] dL/dp (caller scatters it through the max-pool indices and adds it to dx_direct)
 *      du_ws     (T) [B,C,H,W]    workspace (dL/d upsampled map) between the two kernels
 */
int tamtr_cpam_bwd(const void* gout, const void* x, const void* p, const float* s2, const int32_t* arg, void* dx_direct,
                   void* du_ws, void* dp, int B, int C, int H, int W, int dtype, void* stream);

/*      The whole op on channels-last maps (what the trunk runs), its max-pool included: x, p, out, gout, dx and the workspaces (T)
 *      [B][H][W][C] / [B][H/2][W/2][C]; code u8 [B][H/2][W/2][C]: which element of its 3 x 3 window each pooled value is (row-major position in
 *      the unclipped window; first maximum wins, NaN wins - csrc/pool.hip's rules); s2 f32 / arg i32 [B][H][W][8] (chunk innermost).
 *      A lane owns 16 bytes of channels of a pooled cell / a 2 x 2 pixel block / a pixel; the chunk max / sum is a butterfly over the chunk's
 *      lanes: C % 64 == 0 (bf16) / % 32 (f32), C / 8 a power of two times the vector width, 16-byte aligned pointers.  Forward: p, code, out,
 *      s2, arg are written (2 launches).  Backward: gout -> dx (3 launches: gates, bilinear gather into dp_ws, pool gather + direct term);
 *      dxd_ws, du_ws [B][H][W][C] and dp_ws [B][H/2][W/2][C] are caller workspaces.  No transposing copies around the op. */
int tamtr_cpam_cl_fwd(const void* x, void* p, uint8_t* code, void* out, float* s2, int32_t* arg, int B, int C, int H, int W, int dtype, void* stream);
int tamtr_cpam_cl_bwd(const void* gout, const void* x, const void* p, const uint8_t* code, const float* s2, const int32_t* arg, void* dxd_ws,
                      void* du_ws, void* dp_ws, void* dx, int B, int C, int H, int W, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * a-9  SS2D front end: depthwise 3x3 conv + bias + SiLU + cross-scan layout.  Replaces `x = self.act(self.conv2d(x))`
 *      (ultralytics/nn/extra_modules/VManba/vmamba.py:949-952, on the NCHW permutation of the in_proj output) together with
 *      CrossScan (VManba/csms6s.py:4-14) in the pair layout of tamtr_selective_scan_* (xmode = 1).
 *      x      (T) channels-last map: pixel (b, h, w) starts at x + ((b*H + h)*W + w) * x_pixel_stride, D channels are read
 *                 (the in_proj output [B, H, W, 2*d_inner] is passed as it lies, x_pixel_stride = 2*d_inner)
 *      weight f32 [D, 9] (= conv2d.weight [D, 1, 3, 3]), bias f32 [D] or NULL
 *      u2     f32 [B, 2, D, H*W]: plane 0 = SiLU(conv) flattened row-major (h*W + w), plane 1 column-major (w*H + h)
 *      D % 32 == 0; x_pixel_stride a multiple of 16 bytes.
 *      plane_dtype: TAMTR_F32, or TAMTR_BF16 (with dtype = TAMTR_BF16 only): u2 / g2 are bf16 arrays of the same shape ("bf16 PLANES",
 *      tamtr_selective_scan_dtproj_fwd).
 */
int tamtr_dwconv_silu_cross_fwd(const void* x, long long x_pixel_stride, const float* weight, const float* bias, void* u2, int B,
                                int D, int H, int W, int dtype, int plane_dtype, void* stream);
/*      Backward.  g2 f32 [B, 2, D, H*W] (gradient of u2) ->
 *      gx (T) channels-last, pixel stride gx_pixel_stride (D channels written per pixel),
 *      ws f32 [B, tamtr_dwconv_tiles(H, W), D, 10]: per-tile partial sums of d(weight) (9 taps) and d(bias) (caller sums
 *      over the first two axes; no atomics).
 */
int tamtr_dwconv_tiles(int H, int W);
int tamtr_dwconv_silu_cross_bwd(const void* g2, const void* x, long long x_pixel_stride, const float* weight, const float* bias,
                                void* gx, long long gx_pixel_stride, float* ws, int B, int D, int H, int W, int dtype, int plane_dtype,
                                void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * a-9  SS2D back end.  tamtr_cross_merge_* replace CrossMerge (VManba/csms6s.py:26-34) and the transposition to channels-last:
 *          ymT[b, h*W+w, d] = y4[b,0,d,h*W+w] + y4[b,2,d,h*W+w] + y4[b,1,d,w*H+h] + y4[b,3,d,w*H+h]
 *      y4 f32 [B, 4, D, H*W] (scan outputs, un-reversed, as tamtr_selective_scan_* with xmode = 1 write them), ymT f32 [B, H*W, D];
 *      backward: gymT f32 [B, H*W, D] -> g2 f32 [B, 2, D, H*W], the gradient in pair layout that the scan backward reads with
 *      xmode = 3.  D % 32 == 0.  plane_dtype: the element type of y4 / g2 (TAMTR_F32 | TAMTR_BF16, "bf16 PLANES"); ymT / gymT stay f32.
 */
int tamtr_cross_merge_fwd(const void* y4, float* ymT, int B, int D, int H, int W, int plane_dtype, void* stream);
int tamtr_cross_merge_bwd(const float* gymT, void* g2, int B, int D, int H, int W, int plane_dtype, void* stream);

/*      tamtr_ln_gate_* replace `y = self.out_norm(y); y = y * self.act(z)` (VManba/vmamba.py:1005-1008,1029-1036):
 *          out[t, :] = LayerNorm(x[t, :]; gamma, beta, eps) * SiLU(z[t, :]),  z[t, d] = xz[t * xz_token_stride + D + d]
 *      x f32 [ntok, D]; xz (T) the channels-last in_proj output (z = second half of each token's 2*D row); out (T) [ntok, D];
 *      stats f32 [ntok, 2] (mean, 1/std) saved for the backward.  D in {64, 128, 256, 512, 1024}.
 *      Backward: gout (T) [ntok, D] -> gx f32 [ntok, D]; d(z) written into gxz (T) at the same offsets as z in xz (the
 *      caller zero-fills the x half); partials f32 [tamtr_ln_gate_blocks(ntok), 2, D]: per-workgroup sums of d(gamma), d(beta).
 */
int tamtr_ln_gate_blocks(long long ntok);
int tamtr_ln_gate_fwd(const float* x, const void* xz, long long xz_token_stride, const float* gamma, const float* beta, void* out,
                      float* stats, long long ntok, int D, float eps, int dtype, void* stream);
int tamtr_ln_gate_bwd(const void* gout, const float* x, const void* xz, long long xz_token_stride, const float* gamma,
                      const float* beta, const float* stats, float* gx, void* gxz, float* partials, long long ntok, int D, int dtype,
                      void* stream);

/*      Gradient of the two stored cross-scan copies (SS2D backward): out[b, i] = g4[b, i] + g4[b, i + 2] + m_i[b], i = 0, 1.
 *      g4 f32 [B, 4, n] = d/d(u) of the four directions as tamtr_selective_scan_dtproj_bwd writes them (un-reversed), m0 / m1 (T)
 *      [B, n] = the x_proj backward products of the two copies (vmamba.py:962-970), out f32 [B, 2, n], n = D * L, n % 4 == 0.
 */
/*      out = src[0] + ... + src[n-1] (n <= 8 tensors of n_elems elements, T; src: HOST array of device pointers): the gradient of a
 *      tensor with several consumers - the MEH token memory feeds enc_output and every decoder layer's value_proj (head.py:1152-1160,
 *      transformer.py:273) - in one pass instead of autograd's n - 1 pairwise adds. */
int tamtr_sum_n(const void* const* src, int n, void* out, long long n_elems, int dtype, void* stream);

/*      Column sums of a tall bf16 matrix: the bias gradient of a token-wise nn.Linear (value_proj transformer.py:273, enc_output
 *      head.py:1118: db = sum over the B*L tokens of dY).  X bf16 [M, N] -> partial f32 [tamtr_colsum_blocks(M), N], one row of sums per
 *      workgroup, added by the caller (fixed order).  N % 8 == 0, N <= 2048, 256 % (N / 8) == 0. */
int tamtr_colsum_blocks(long long M);
int tamtr_colsum_bf16(const void* X, float* partial, long long M, int N, void* stream);
/*      out f32 [C] = sum over the R rows of in (T) [R, C], added in a fixed order in ONE launch: the last stage of the package's two-stage
 *      reductions - the d(gamma) / d(beta) partial rows of LayerNorm (vmamba.py:1190,1222 backward), the depthwise convolution's
 *      d(weight) tile sums (vmamba.py:949-952), the scan's per-image d(A) / d(D) rows (csms6s.py:267), the row-slice products of the
 *      tall linears' weight gradients (vmamba.py:935,1010 in_proj / out_proj).  Stands where `partials.sum(0)` stood: torch's
 *      multi-workgroup reduction zeroes a semaphore buffer with a memset node per launch, which a HIP-graph replay under AQL packet
 *      capture does not keep in order with the kernels around it.  C % 4 == 0; in 16-byte (f32) / 8-byte (bf16) aligned. */
int tamtr_slab_sum_rows(const void* in, float* out, int R, long long C, int dtype, void* stream);
int tamtr_fold_add(const float* g4, const void* m0, const void* m1, float* out, int B, long long n, int dtype, void* stream);

/*      tamtr_layernorm_* : LayerNorm over the channel axis of a token-major map, VSSBlock.norm / norm2
 *      (VManba/vmamba.py:1190,1222,1239-1254).  x, out, gout, gx (T) [ntok, D]; gamma, beta f32 [D]; stats f32 [ntok, 2];
 *      partials f32 [tamtr_ln_gate_blocks(ntok), 2, D] (per-workgroup d(gamma), d(beta) sums).  D in {32,...,1024} powers of two.
 */
int tamtr_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* out, float* stats, long long ntok, int D,
                        float eps, int dtype, void* stream);
int tamtr_layernorm_bwd(const void* gout, const void* x, const float* gamma, const float* stats, void* gx, float* partials,
                        long long ntok, int D, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * trunk  Training-mode BatchNorm2d (+ SiLU).  Replaces `self.act(self.bn(y))` of Conv.forward, ultralytics/nn/modules/conv.py:36-40,
 *      in train mode (batch statistics, running-stat update with the unbiased variance, as nn.BatchNorm2d).
 *      x, y, gy, gx (T) [B, C, HW] (NCHW maps); gamma, beta, running_mean, running_var f32 [C] (running_* may be NULL);
 *      mean_rstd f32 [C, 2] saved for the backward; partials: caller workspace of C * tamtr_bn_slices(B, HW) * 3 floats
 *      (forward) / * 2 floats (backward).  act: 0 = identity, 1 = SiLU.  Backward returns d(gamma), d(beta) f32 [C].
 */
int tamtr_bn_slices(int B, int HW);
int tamtr_bn_act_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, void* y,
                     float* mean_rstd, float* partials, int B, int C, int HW, float eps, float momentum, int act, int dtype, void* stream);
int tamtr_bn_act_bwd(const void* gy, const void* x, const float* gamma, const float* beta, const float* mean_rstd, void* gx,
                     float* ggamma, float* gbeta, float* partials, int B, int C, int HW, int act, int dtype, void* stream);

/*      Channels-last variant (token-major maps, e.g. the MEH input projection output [B*L, hd] of head.py:1087 run as a GEMM):
 *      x, y, gy, gx (T) [N, C], C contiguous and 16-byte aligned, C <= 1024, C % 4 == 0 and 256 % (C / V) == 0 with V = 8 for
 *      bf16 maps with C % 8 == 0, else 4.  partials: caller workspace of S = tamtr_bncl_blocks(N, C, dtype) chunks:
 *      C * S * 3 floats (forward) / C * S * 2 + 2 * C floats (backward).  ldgy: row pitch of gy in elements (C = packed; larger when
 *      the gradient is a channel slice of the gradient of a `torch.cat(..., 1)`, so that no packing copy is needed).
 */
int tamtr_bncl_blocks(long long N, int C, int dtype);
int tamtr_bncl_act_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, const void* residual,
                       void* y, float* mean_rstd, float* partials, long long N, int C, float eps, float momentum, int act, int dtype,
                       void* stream);
/*      residual (T [N, C], may be NULL): y = act(bn(x)) + residual - the shortcut of RepNBottleneck (`x + self.cv2(self.cv1(x))`,
 *      extra_modules/block.py:100-102) joins in the apply pass; its gradient is gy itself (the caller passes it on).
 *      tamtr_bncl2_act_*: y = act(bn1(x1) + bn2(x2)), both in training mode over the same shape - RepConvN's training form
 *      `self.act(self.conv1(x) + self.conv2(x))` (block.py:66-69).  mean_rstd f32 [2][C][2]; partials 2 * C * S * 3 floats forward,
 *      C * S * 3 + 3 * C backward.
 */
int tamtr_bncl_stats(const void* x, float* running_mean, float* running_var, float* mean_rstd, float* partials, long long N, int C, float eps,
                     float momentum, int dtype, void* stream);   /* batch statistics + running update only: mean_rstd f32 [C][2] */
int tamtr_bncl2_act_fwd(const void* x1, const float* gamma1, const float* beta1, float* running_mean1, float* running_var1, const void* x2,
                        const float* gamma2, const float* beta2, float* running_mean2, float* running_var2, void* y, float* mean_rstd,
                        float* partials, long long N, int C, float eps, float momentum, int act, int dtype, void* stream);
int tamtr_bncl2_act_bwd(const void* gy, long long ldgy, const void* x1, const void* x2, const float* gamma1, const float* beta1,
                        const float* gamma2, const float* beta2, const float* mean_rstd, void* gx1, void* gx2, float* ggamma1, float* gbeta1,
                        float* ggamma2, float* gbeta2, float* partials, long long N, int C, int act, int dtype, void* stream);
int tamtr_bncl_act_bwd(const void* gy, long long ldgy, const void* x, const float* gamma, const float* beta, const float* mean_rstd,
                       void* gx, float* ggamma, float* gbeta, float* partials, long long N, int C, int act, int dtype, void* stream);

/* ---- layout: NCHW <-> NHWC repacking of a feature map (tiled transpose through LDS).  The trunk runs channels-last around
 *      MIOpen's NHWC convolutions; the gate (a-1) and CPAM (next-3) kernels read NCHW planes.  Replaces torch's
 *      `.contiguous()` / `.contiguous(memory_format=torch.channels_last)` on those edges (same values, no arithmetic).
 *      to_nhwc != 0: src [B, C, HW] -> dst [B, HW, C];  to_nhwc == 0: src [B, HW, C] -> dst [B, C, HW].  T = f32 | bf16.
 *      ld: pixel pitch of the NHWC side in elements (ld == C: packed; ld > C: a channel slice of a wider map, e.g. one half of
 *      `cv1(x).chunk(2, 1)`, extra_modules/block.py:147).
 */
int tamtr_relayout(const void* src, void* dst, int B, int C, int HW, int ld, int to_nhwc, int dtype, void* stream);
/*      Channel-slice copy on channels-last maps: dst[r][0..C) = src[r][0..C), N = B*H*W rows, row pitches lds / ldd elements.
 *      Replaces the strided copies behind `torch.cat(y, 1)` / `x.chunk(2, 1)` (extra_modules/block.py:133,147,152) in NHWC. */
/*      Nearest-neighbour x2 / x0.5 resampling of a packed NHWC map = nn.Upsample(scale_factor=2.0 | 0.5, mode='nearest') as TAMTR.yaml
 *      uses it, forward and backward.  mode 0: dst [B,2H,2W,C] <- src [B,H,W,C] (up);  1: dst [B,H,W,C] <- src [B,2H,2W,C] (its
 *      gradient: sum of the four);  2: dst [B,H/2,W/2,C] <- src [B,H,W,C] (down: every second pixel);  3: dst [B,H,W,C] <- src
 *      [B,H/2,W/2,C] (its gradient: scattered, zeros elsewhere, one pass).  C % 8 == 0 (bf16) / C % 4 == 0 (f32). */
int tamtr_resample2(const void* src, void* dst, int B, int H, int W, int C, int mode, int dtype, void* stream);
int tamtr_copy_rows(const void* src, long long lds, void* dst, long long ldd, long long N, int C, int dtype, void* stream);
/*      The whole `torch.cat(y, 1)` of up to four channels-last maps in one launch: dst[r][off_k .. off_k + C[k]) = src[k][r][0 .. C[k]),
 *      off_k = C[0] + ... + C[k-1]; src / lds / C are HOST arrays of n <= 4 entries (pointers are device pointers); widths, pitches
 *      and pointers 16-byte granular, else TAMTR_EUNSUP (then copy input by input). */
int tamtr_cat_rows(const void* const* src, const long long* lds, const int* C, int n, void* dst, long long ldd, long long N, int dtype,
                   void* stream);

/* ---- max pooling k x k / stride s / padding p (floor mode), NCHW (nhwc = 0) or NHWC (nhwc = 1) maps.  Replaces nn.MaxPool2d as
 *      used by SPPELAN (5/1/2, three chained: ultralytics/nn/extra_modules/block.py:255-268) and by CPAM's channel gate (3/2/1,
 *      block.py:274), forward and backward.  code: one byte per output element = position of the winner inside the unclipped
 *      window (row-major; first maximum wins, NaN wins over everything - torch's rule).  The backward gathers (no atomics).
 *      addend (T, same layout as gx, may be NULL): gx = addend + pooled gradient (CPAM adds its direct term this way).
 *      x, gx (T) [B,C,H,W] in the given layout; y, gy (T) and code (u8) [B,C,Ho,Wo], Ho = tamtr_maxpool_out(H, k, s, p).  k <= 15.
 */
int tamtr_maxpool_out(int n, int k, int s, int p);
int tamtr_maxpool_fwd(const void* x, void* y, uint8_t* code, int B, int C, int H, int W, int k, int s, int p, int nhwc, int dtype,
                      void* stream);
int tamtr_maxpool_bwd(const void* gy, const uint8_t* code, const void* addend, void* gx, int B, int C, int H, int W, int k, int s, int p, int nhwc, int dtype,
                      void* stream);

/* ---- data path: the pixel half of the training transforms on the device.
 *      Replaces, for a whole batch and in this order: cv2.warpAffine(img, M[:2], dsize, borderValue=114) of RandomPerspective
 *      (ultralytics/data/augment.py:415-420), the BGR2HSV -> LUT -> HSV2BGR chain of RandomHSV (:590-609), RandomFlip's
 *      np.flipud / np.fliplr (:656-660), Format._format_img (:920-926) and preprocess_batch's img.float() / 255
 *      (models/yolo/detect/train.py:56).  Same arithmetic as libtamtr_host.so (include/tamtr_host.h), bit for bit.
 *      src        u8  [B, SH, SW, 3]  decoded + stretched RGB images (before any transform)
 *      inv_affine f64 [B, 6]          destination -> source map  m0 m1 b1 / m3 m4 b2  (inverse of the transform's 2x3 matrix)
 *      luts       u8  [B, 3, 256]     hue (< 180), saturation, value look-up tables of RandomHSV
 *      flags      i32 [B]             bit 0: flipped up-down, bit 1: flipped left-right, bit 2: no HSV step (luts ignored)
 *      out        f32 [B, 3, H, W]    pixel * float(1 / 255.0), which is how torch evaluates `img.float() / 255` on the device
 */
int tamtr_img_augment_u8(const uint8_t* src, const double* inv_affine, const uint8_t* luts, const int32_t* flags, float* out, int B,
                         int SH, int SW, int H, int W, int border, void* stream);

/*      The reference's optimizer_step (ultralytics/engine/trainer.py:471-479): clip_grad_norm_(max_norm) -> AdamW.step() ->
 *      ModelEMA.update (utils/torch_utils.py:392-419), over a device table of tensors built once by the caller; torch's FusedAdamKernel
 *      arithmetic in fp32.  All table arrays are DEVICE arrays with one entry per tensor unless noted:
 *        p, m, v, e   addresses of the values, exp_avg, exp_avg_sq and EMA copy (m[i] == 0: EMA-only entry, e.g. BatchNorm statistics;
 *                     e[i] == 0: no EMA), all f32 with the element order of p;  step f32: Adam step counts (advanced for tensors that have a
 *                     gradient);  numel i64;  group u8: parameter group;  chunk_tensor i32 / chunk_off i64 [nchunks]: the tensors cut into
 *                     pieces of tamtr_optim_chunk() elements;  grads: THIS step's gradient addresses (0: none - Adam skipped, EMA still taken).
 *        partial f32 [nchunks], normcoef f32 [2] workspaces; normcoef = {total gradient norm, clip coefficient} afterwards (device side:
 *                     nothing is read back).  lr, wd: HOST arrays [ngroups <= 4].  max_norm <= 0: no clipping.  do_ema == 0: EMA untouched.
 *      The clipped gradient is written back to the gradient buffers, as clip_grad_norm_ leaves it.
 *        shadow       NULL, or a device array of addresses of bf16 arrays with the element order of p (0: none): the updated value of every
 *                     tensor that has a gradient this step is ALSO stored there rounded to bf16 - the compute copy that bf16 autocast casts out
 *                     of the fp32 master at every use (trainer.py:343 `torch.cuda.amp.autocast`); engine.FusedOptimStep(shadows=True). */
int tamtr_optim_chunk(void);
int tamtr_optim_step(const void* const* p, const void* const* m, const void* const* v, const void* const* e, const void* const* shadow, float* step,
                     const long long* numel,
                     const unsigned char* group, const int* chunk_tensor, const long long* chunk_off, const void* const* grads, int ntensors,
                     int nchunks, float* partial, float* normcoef, const float* lr, const float* wd, int ngroups, float beta1, float beta2, float eps,
                     float max_norm, float ema_decay, int do_ema, void* stream);

/*      x_proj of SS2D on the cross-scan pair layout (ultralytics/nn/extra_modules/VManba/vmamba.py:962-975: x_dbl = einsum("b k d l, k c d
 *      -> b k c l", xs, x_proj_weight); dts, Bs, Cs = split(x_dbl, [R, N, N])), bf16 MFMA products on f32 operands as the scan kernels keep
 *      them (values rounded to bf16 in registers, products rounded to bf16 before they are stored / added: the arithmetic of the bf16
 *      library GEMMs these kernels replace).  C = R + 32, N = 16 states, 1 <= R <= 32, 2C in {66 .. 128}.
 *        u2    f32 [B, 2, D, L]     SiLU(dwconv(x)) row-major / column-major (tamtr_dwconv_silu_cross_fwd); directions k, k + 2 read copy k & 1
 *        wcat  bf16 [2, MP, D]      rows [W_i ; W_(i+2)] of x_proj_weight [4, C, D], zero-padded to MP = ceil(2C / 32) * 32 rows
 *        dtr   f32 [B, 4, R, L], Bs, Cs f32 [B, 4, 16, L]   un-reversed, as tamtr_selective_scan_dtproj_* take them
 *      backward:  gdtr, gB, gC f32 as the scan backward writes them; gu f32 [B, 4, D, L] = the scan's d/d(u) per direction;
 *        wT    bf16 [2, D, KP]      wcat transposed, zero-padded to KP = ceil(2C / 16) * 16 columns
 *        gu2   f32 [B, 2, D, L]   = gu[:, i] + gu[:, i + 2] + Wcat_i^T G_i: the gradient of the two stored copies (what tamtr_fold_add + the
 *                                   product did in two passes)
 *        part  f32 [B * tamtr_xproj_dw_slices(L), 2, 2C, D]   per-(image, 1 024-pixel slice) partial dWcat tiles; the caller adds them in
 *                                   order (tamtr_slab_sum_rows) and re-splits rows [0, C) / [C, 2C) of copy i into directions i / i + 2
 *      D % 16 == 0 (fwd), % 32 (dx), % 256 and L % 8 == 0 (dw); wcat / wT 16-byte aligned.
 *      plane_dtype = TAMTR_BF16 ("bf16 PLANES"): u2, gu and gu2 are bf16 arrays of the same shapes (L % 2 == 0); a lane then handles a pixel
 *      PAIR (one dword), a wave 64 pixels; dtr / Bs / Cs / gdtr / gB / gC / part stay f32. */
int tamtr_xproj_dw_slices(int L);
int tamtr_xproj_fwd(const void* u2, const void* wcat, float* dtr, float* Bs, float* Cs, int B, int D, int L, int R, int plane_dtype, void* stream);
int tamtr_xproj_bwd_dx(const void* gu, const float* gdtr, const float* gB, const float* gC, const void* wT, void* gu2, int B, int D, int L, int R,
                       int plane_dtype, void* stream);
int tamtr_xproj_bwd_dw(const void* u2, const float* gdtr, const float* gB, const float* gC, float* part, int B, int D, int L, int R, int plane_dtype,
                       void* stream);

/*      The per-layer terms of the RT-DETR loss and the matcher's cost matrix (a-10: ultralytics/models/utils/loss.py:85-166,282-326 - varifocal
 *      class loss on the matched IoU, 5 x L1, 2 x (1 - RIOU); RIOU: ultralytics/utils/metrics.py:91-130; cost: models/utils/ops.py:84-112), all
 *      decoder layers of a call stacked, fp32.  pb f32 [Lr, B, nq, 4] (xywh), ps f32 [Lr, B, nq, nc] (logits), gt_bboxes f32 [G, 4], gt_cls i64 [G];
 *      matched pairs as flat i64 lists li, bi, si, gi [Lr * n] (layer-major, exactly n pairs per layer, a query matched at most once per layer).
 *      fwd: workspaces tgt i64 [Lr, B, nq] PRESET to nc (no object), score f32 [Lr, B, nq] PRESET to 0, pair_l1 / pair_riou f32 [Lr * n],
 *           partial f32 [Lr * tamtr_detr_blocks(B * nq)];  out f32 [3, Lr] = (class, bbox, giou) terms, gains and 1 / n applied.
 *      bwd: up f32 [3, Lr] upstream gradients of `out`; gps f32 [Lr, B, nq, nc] (every element written), gpb f32 [Lr, B, nq, 4] (the matched rows
 *           written: the caller zero-fills).  RIOU's alpha is a constant (torch.no_grad in the reference); ties of max / min split the gradient.
 *      match_cost: C f32 [rows = Lr * B * nq, G] = g_class (focal pos - neg) + g_bbox L1 + g_giou (1 - RIOU), non-finite -> 0. */
int tamtr_detr_blocks(int rows_per_layer);
int tamtr_detr_layers_fwd(const float* pb, const float* ps, const float* gt_bboxes, const long long* gt_cls, const long long* li, const long long* bi,
                          const long long* si, const long long* gi, int Lr, int B, int nq, int nc, int n, long long* tgt, float* score, float* pair_l1,
                          float* pair_riou, float* partial, float g_class, float g_bbox, float g_giou, float* out, void* stream);
int tamtr_detr_layers_bwd(const float* pb, const float* ps, const float* gt_bboxes, const long long* li, const long long* bi, const long long* si,
                          const long long* gi, const long long* tgt, const float* score, const float* up, int Lr, int B, int nq, int nc, int n,
                          float g_class, float g_bbox, float g_giou, float* gpb, float* gps, void* stream);
int tamtr_detr_match_cost(const float* ps, const float* pb, const float* gt_bboxes, const long long* gt_cls, long long rows, int nc, int G, float g_class,
                          float g_bbox, float g_giou, float alpha, float gamma, float* C, void* stream);

/*      tamtr_bncl_act_fwd / _bwd with the OUTPUT (forward) / the incoming GRADIENT (backward) laid out as image segments of a wider buffer:
 *      level i of the MEH token memory `feats` [B, L_0 + L_1 + L_2, hd] is the BatchNorm of its input projection written straight into
 *      feats[:, off_i : off_i + L_i] (ultralytics/nn/modules/head.py:1087,1202-1219: `input_proj[i]` ... `torch.cat(feats, 1)`), and the
 *      backward reads d(feats)[:, off_i : off_i + L_i] where it lies - no concatenation copy, no slice copies.
 *      x (T) [N = B * seg_rows, C] packed; y / gy point at image 0's first row of the segment, image b's rows start b * seg_pitch elements
 *      further (seg_pitch = (L_0 + L_1 + L_2) * C); C a power of two; everything else as tamtr_bncl_act_fwd / _bwd. */
int tamtr_bncl_act_seg_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, void* y, long long seg_rows,
                           long long seg_pitch, float* mean_rstd, float* partials, long long N, int C, float eps, float momentum, int act, int dtype,
                           void* stream);
int tamtr_bncl_act_seg_bwd(const void* gy, long long seg_rows, long long seg_pitch, const void* x, const float* gamma, const float* beta,
                           const float* mean_rstd, void* gx, float* ggamma, float* gbeta, float* partials, long long N, int C, int act, int dtype,
                           void* stream);

/*      The decoder's iterative box refinement (ultralytics/nn/modules/transformer.py:881-887, nn/modules/utils.py:46-52):
 *          out = sigmoid(delta + inverse_sigmoid(ref)),  inverse_sigmoid(x) = log(max(clamp(x, 0, 1), 1e-5) / max(1 - clamp(x, 0, 1), 1e-5))
 *      delta, ref, out f32 [n] (any shape, contiguous).  Backward: gdelta = gout out (1 - out); gref (NULL: not wanted - the reference detaches
 *      `ref` between layers) = gdelta * d(inverse_sigmoid)/d(ref) with torch's clamp gradients. */
int tamtr_box_refine_fwd(const float* delta, const float* ref, float* out, long long n, void* stream);
int tamtr_box_refine_bwd(const float* gout, const float* out, const float* ref, float* gdelta, float* gref, long long n, void* stream);

/*      Node census of the graph that `stream` is capturing into (hipStreamGetCaptureInfo_v2 + hipGraphGetNodes): counts[t] = nodes of
 *      hipGraphNodeType t, t < n_types <= 16 (0 kernel, 1 memcpy, 2 memset, ...).  Host-side helper of the HIP-graph replay
 *      (tam-tr_amd/graphs.py: memset nodes do not survive AQL packet capture); TAMTR_EINVAL when the stream is not capturing. */
int tamtr_graph_capture_census(void* stream, int* counts, int n_types);

#ifdef __cplusplus
}
#endif
#endif /* TAMTR_HIP_H */
