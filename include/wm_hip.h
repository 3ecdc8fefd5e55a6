/* wm_hip.h -- C ABI of libwm_hip.so: the MI355X (gfx950) kernels of the watermark embed+detect hot path.
 *
 * The reference (py/main16.py) has no FFI: its boundary for this path is the nn.Module call
 *   Generator.forward(s, message)  py/main16.py:149-162
 *   Detector.forward(x)            py/main16.py:183-186
 * plus the free functions / loss modules of py/main16.py:53-81 and :192-217.  Everything those calls reach in
 * ATen (conv1d, batch_norm, lstm, conv_transpose1d, embedding, stft, ...) is replaced by the stateless launchers
 * below; the host-side mirror of the nn.Module API lives in the Python package and binds this file with ctypes
 * (see INTEGRATION.md for the binding a maintainer of the reference would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless stated (message: int64, tables: int32);
 *   - activations are channel-first frames [B,64,T] exactly as the reference holds them, T % 4 == 0;
 *   - inputs are borrowed and never written; outputs / scratch are caller-allocated;
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); launchers enqueue and return,
 *     they never synchronise, allocate or free (graph-capture safe).  State kept across calls: (a) a per-device
 *     cache of "dynamic-LDS opt-in done" bits (hipFuncSetAttribute is per device; idempotent, so launchers are
 *     safe from several host threads and for one process driving several GPUs); (b) the two PROCESS-WIDE
 *     experiment knobs wm_set_conv_bf_schedule / wm_set_lstm_dx_bf16x6 below -- they select between kernel
 *     generations of equal results, are meant to be set once at start-up (or never: the defaults are the fast
 *     builds) and are not synchronised with launches issued concurrently from other threads;
 *   - return value: 0 on success, else a hipError_t value (1 = invalid argument / unsupported variant).
 */
#ifndef WM_HIP_H
#define WM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* wm_stream_t;   /* == hipStream_t */

/* ---- 64->64 convolutions on the fp32 matrix cores -------------------------------------------------------------
 * replace nn.Conv1d(64,64,3,padding=1) of ResBlock (py/main16.py:116,119) and nn.ConvTranspose1d(64,64,7,padding=3)
 * (py/main16.py:144), forward and backward.                                                                     */

/* w -> GEMM image wp[KW][64 in][64 out].  mode 0 Conv1d fwd | 1 Conv1d dgrad | 2 ConvTranspose1d fwd | 3 ConvT dgrad */
int wm_pack_w64(const float* w, float* wp, int KW, int mode, wm_stream_t stream);

/* y = conv_same(pro(x)) then epi.
 *   pro: 0 x | 1 relu(x*pa[c]+pb[c]) (BatchNorm+ReLU folded into the load, py/main16.py:117-118)
 *        | 2 x + pa[b*64+c] (message embedding add, py/main16.py:158-159) | 3 pa[c]*x + (pb[c] + pb[64+c]) + pc[c]*x2 (BN backward; pb is [2][64]: offset as hi + lo words)
 *   epi: 0 + bias[c] | 1 keep where e1*ea[c]+eb[c] > 0 (ReLU backward) | 2 + e1 (residual gradient) | 3 none
 *   stats (NULL or [256][2][64]): per-workgroup partial sums for BatchNorm (epi 0: sum y, sum y^2;
 *   epi 1: sum v, sum v*e1).  Supported (KW,pro,epi): (3,{0,1},0) (3,3,{1,2,3}) (7,{0,2},0) (7,0,3).          */
int wm_conv64(const float* x, const float* x2, const float* wp, const float* pa, const float* pb, const float* pc,
              const float* bias, const float* e1, const float* ea, const float* eb, float* y, float* stats,
              int B, int T, int KW, int pro, int epi, wm_stream_t stream);

/* bf16x6 build of the k3 convolution: every fp32 operand is split into three bf16 pieces in LDS and the six piece
 * products of weight >= 2^-16 are accumulated in fp32 on the bf16 matrix cores (fp32-grade error, 6/16 of the fp32 MFMA
 * time).  Same pro / epi / stats contract as wm_conv64 with KW = 3; wpb [3][3][64][64] uint16 from wm_pack_w64_bf
 * (mode 0 Conv1d fwd | 1 Conv1d dgrad).  One more epilogue exists here only, for inference (py/main16.py:124-125 in
 * eval mode as two launches): (pro 1, epi 4, stats NULL) y = relu(e1 + (conv + bias[c]) * ea[c] + eb[c]), i.e. the
 * second conv of a ResBlock with BatchNorm2 (folded running statistics), the residual add and the ReLU in its epilogue;
 * needs schedule 2 and T % 128 == 0 (hipErrorInvalidValue otherwise).
 * For the 64-channel stride-1 blocks of the main14b_2 variant (py/main14b_2.py:95-102, no BatchNorm; pro 0, stats NULL, any
 * T % 4 == 0, phase-serial kernel): epi 5 y = elu(conv + bias) | 6 y = elu(conv + bias + e1) | 7 y = conv * ELU'(e1), e1 = the ELU
 * output the gradient flows into | 2 y = conv + e1 | 3 y = conv. */
int wm_pack_w64_bf(const float* w, void* wpb, int mode, wm_stream_t stream);
/* schedule of wm_conv64_bf (process-wide knob; default 2): 0 phase-serial, one wave per SIMD, 128-column tiles |
 * 2 weight fragments resident in registers, input image double-buffered, split / deferred epilogue / prefetch dealt out one
 * slice per MFMA (T % 128 == 0, else 0 runs).  Any other value selects 0. */
int wm_set_conv_bf_schedule(int schedule, wm_stream_t stream);
int wm_conv64_bf(const float* x, const float* x2, const void* wpb, const float* pa, const float* pb, const float* pc,
                 const float* bias, const float* e1, const float* ea, const float* eb, float* y, float* stats,
                 int B, int T, int pro, int epi, int arith, wm_stream_t stream);
/* arith 0: bf16x6, wpb from wm_pack_w64_bf.  arith 1: f16 two-piece split, three products per product on v_mfma_f32_32x32x16_f16
 * (see wm_dwgrad64_bf below), wpb from wm_pack_w64_h(mode 0); only (pro 0 | 1, epi 0, with or without stats) and (pro 1, epi 4) under
 * schedule 2 with T % 128 == 0 -- the ResBlock forward convolutions -- hipErrorInvalidValue otherwise.  Activations are split unscaled: the
 * representation floor is 2^-25 ABSOLUTE (f16 subnormal spacing / 2) on top of 2^-22 relative, i.e. fp32-grade for the O(1)
 * activations BatchNorm + ReLU produce. */

/* Inference ResBlock as ONE launch (py/main16.py:112-125 with both BatchNorm1d in eval mode):
 *   y = relu(x + (conv2(relu((conv1(x) + b1) * sc1 + sh1)) + b2) * sc2 + sh2)
 * sc / sh = the folded running statistics (wm_bn_eval_scale_shift); w1pb / w2pb = wm_pack_w64_bf_scaled images of
 * w1 * sc1[out] / w2 * sc2[out] (the per-channel scale rides in the weights, the kernel adds b * sc + sh); b1 / b2 may be NULL.
 * Two frame passes over HBM (x in, y out): the intermediate activation stays in LDS as bf16x3 pieces.  T % 4 == 0. */
int wm_pack_w64_bf_scaled(const float* w, const float* row_scale, void* wpb, wm_stream_t stream);
int wm_resblock_eval_bf(const float* x, const void* w1pb, const void* w2pb, const float* b1, const float* sc1, const float* sh1,
                        const float* b2, const float* sc2, const float* sh2, float* y, int B, int T, int arith, wm_stream_t stream);
/* arith 0: bf16x6 (images from wm_pack_w64_bf_scaled).  arith 1: f16 two-piece split, three products per product; images from
 * wm_pack_w64_h_scaled (w * sc[out] * ws, {ws, 1 / ws} behind the image), x and the intermediate split unscaled. */
int wm_pack_w64_h_scaled(const float* w, const float* row_scale, void* wph, wm_stream_t stream);

/* Data gradient AND weight gradient of a 64->64 k3 convolution in ONE launch (ResBlock backward, py/main16.py:112-125 under
 * autograd): g = ga[c] dz + gb[c] + gb[64+c] + gc[c] y is rebuilt once and feeds both; frames moved: 4 (conv2 pair) / 5 (conv1
 * pair) instead of 7.  wpb = wm_pack_w64_bf mode-1 image.  Two forms:
 *   xpro 1, epi 1: x' = relu(x xa + xb) is the weight gradient's input operand; y = data gradient masked by (e1 ea + eb > 0),
 *                  stats [256][2][64] = (sum y, sum y e1) per workgroup (reduce with wm_bn_bwd_finalize)       [conv2 of a block]
 *   xpro 0, epi 2: x as is; y = data gradient + e1; stats NULL                                                 [conv1 of a block]
 *   xpro 0, epi 8: the second form for a block that FOLLOWS another ResBlock: y = (data gradient + e1) masked by the sign bits of
 *                  the previous block's output (eb = that mask, passed as const float*), stats = (sum y, sum y ea) with ea = the
 *                  previous block's pre-BatchNorm activation y2 [B,64,T]: the previous block's ReLU backward and BatchNorm sums
 *                  ride in this launch's epilogue (it then needs neither wm_relu_bwd_reduce_mask nor a mask of its own); needs gmask.
 * gmask (optional): the sign bits wm_bn_add_relu_mask wrote for the block output.  With it the gradient that reaches the block
 * output is passed as it arrived and masked on load -- as dz in the first form, as e1 in the second -- so the masked copy
 * dz = g (out > 0) never exists in memory (wm_relu_bwd_reduce_mask with dz = NULL supplies the two BatchNorm sums).
 * dw [out][in][3] / dbias [64] as wm_wgrad64_bf (partial: 256 x (3*4096+64) floats of scratch; accumulate 0 | 1).  T % 64 == 0. */
int wm_dwgrad64_bf(const float* g, const float* g2, const float* ga, const float* gb, const float* gc, const void* wpb,
                   const float* x, const float* xa, const float* xb, const float* e1, const float* ea, const float* eb,
                   float* y, float* stats, float* partial, float* dw, float* dbias, int B, int T, int xpro, int epi, int accumulate,
                   const void* gmask, int arith, const float* gscale, float* dzmax, wm_stream_t stream);
/* arith 0: bf16 three-piece split, six piece products (bf16x6); wpb from wm_pack_w64_bf.
 * arith 1: f16 TWO-piece split (22 bits per operand), three products on v_mfma_f32_32x32x16_f16 -- half the matrix work of arith 0,
 *          wpb from wm_pack_w64_h (weights scaled by a
 *          power of two chosen from max |w|, stored behind the image); gscale = wm_bn_bwd_finalize's {gs, 1 / gs} for THIS launch's
 *          g; dzmax (optional, epi 1 / 2 / 8): 256 floats, max |y| per workgroup = the dzmax input of the next launch's finalize. */
int wm_pack_w64_h(const float* w, void* wph, int mode, wm_stream_t stream);      /* 2 * 3 * 4096 f16 + 2 floats */

/* bf16x6 build of the 7-tap ConvTranspose1d(64,64,7,padding=3) (py/main16.py:144): wpb [3][7][64][64] uint16 from
 * wm_pack_w64_bf7 (mode 2 forward | 3 data gradient).  pro 0 x | 2 x + vec[b*64+c]; epi 0 + bias[c] | 3 none. */
int wm_pack_w64_bf7(const float* w, void* wpb, int mode, wm_stream_t stream);
int wm_conv64_bf7(const float* x, const void* wpb, const float* vec, const float* bias, float* y, int B, int T, int pro, int epi,
                  int arith, const float* gscale, wm_stream_t stream);
/* arith 0: bf16x6, wpb from wm_pack_w64_bf7.  arith 1 (T % 128 == 0): f16 two-piece split, three products per product, wpb
 * [2][7][64][64] f16 + {ws, 1 / ws} from wm_pack_w64_h7 (same modes); gscale (optional; the data-gradient launch passes it) = {gs, 1 / gs}
 * from wm_gscale_absmax: the input is multiplied by gs before the split and clamped to +-6e4, the result leaves times 1 / (ws gs). */
int wm_pack_w64_h7(const float* w, void* wph, int mode, wm_stream_t stream);     /* 2 * 7 * 4096 f16 + 2 floats */
/* {gs, 1 / gs} for a gradient tensor x [n] (n % 4 == 0): gs = the power of two that puts max |x| into (2^(L-1), 2^L], L = log2_target
 * (12 leaves 2^4 of headroom below the f16 maximum); all-zero / non-finite input gives gs = 1.  scratch >= 1024 floats. */
int wm_gscale_absmax(const float* x, long long n, float* scratch, float log2_target, float* gscale, wm_stream_t stream);
/* the same from per-workgroup maxima a producer already wrote (wm_dwgrad64_bf's dzmax, epi 1 / 2 / 8): no pass over the tensor */
int wm_gscale_from_max(const float* maxes, int n, float log2_target, float* gscale, wm_stream_t stream);

/* weight gradient of that ConvTranspose1d (= wm_wgrad64 with KW 7, gpro 0, layout 1): dw [in][out][7], dbias [64];
 * xpro 0 | 2 (x + vec[b*64+c]); partial: >= 256 * (7*4096 + 64) floats. */
int wm_wgrad64_bf7(const float* g, const float* x, const float* vec, float* partial, float* dw, float* dbias, int B, int T,
                   int xpro, int accumulate, int arith, const float* gscale, wm_stream_t stream);
/* arith 1: f16 two-piece split; gscale = wm_gscale_absmax's {gs, 1 / gs} for g (required), x is split unscaled, dbias from the unscaled g */

/* bf16x6 build of the k3 Conv1d weight gradient (contract of wm_wgrad64 with KW = 3, layout 0; gpro / xpro (3,1), (3,0), (0,0)).  accumulate: bit 0 = add to
 * dw / dbias, bit 1 = the output-split build (a wave keeps one 32x32 block per tap: ~200 registers per lane, so the workgroup
 * can share a CU with the LSTM recurrence kernels when it is launched on a side stream; same results)                     */
int wm_wgrad64_bf(const float* g, const float* g2, const float* ga, const float* gb, const float* gc,
                  const float* x, const float* xa, const float* xb, float* partial, float* dw, float* dbias,
                  int B, int T, int gpro, int xpro, int accumulate, wm_stream_t stream);

/* dW (+)= sum_{b,t} gpro(g)[out,t] * xpro(x)[in,t+tap-KW/2]; dbias (+)= sum gpro(g).  partial: [512][KW*4096+64].
 *   gpro 0|3, xpro 0|1|2 as above; layout 0: Conv1d weight [out][in][KW], 1: ConvTranspose1d weight [in][out][KW]. */
int wm_wgrad64(const float* g, const float* g2, const float* ga, const float* gb, const float* gc,
               const float* x, const float* xa, const float* xb, float* partial, float* dw, float* dbias,
               int B, int T, int KW, int gpro, int xpro, int layout, int accumulate, wm_stream_t stream);

/* ---- BatchNorm1d(64) glue (py/main16.py:117,120) ----------------------------------------------------------- */
int wm_bn_finalize(const float* partials, int nparts, double count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
                   float* scale, float* shift, float* save_mean, float* save_invstd, wm_stream_t stream);
int wm_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                           float eps, float* scale, float* shift, wm_stream_t stream);
/* ResBlock tail  out = relu(x + y2*scale + shift)  (py/main16.py:125) and its backward + BN-backward reductions */
int wm_bn_add_relu(const float* x, const float* y2, const float* scale, const float* shift, float* out, int B, int T,
                   wm_stream_t stream);
int wm_relu_bwd_reduce(const float* g, const float* out, const float* y2, float* dz, float* partial, int B, int T,
                       wm_stream_t stream);
/* The same pair with the sign of `out` carried as one bit per element: mask = ceil(T / 32) 32-bit words per (clip, channel) row,
 * B * 64 rows, bit (t % 32) of word t / 32 set where out > 0.  The backward then reads g, y2 and 3 % of a frame instead of g, out,
 * y2; with dz = NULL it only forms the two sums (the masked gradient is then rebuilt on load by wm_dwgrad64_bf's gmask).  What
 * ResBlock training uses. */
int wm_bn_add_relu_mask(const float* x, const float* y2, const float* scale, const float* shift, float* out, void* mask, int B, int T,
                        wm_stream_t stream);
int wm_relu_bwd_reduce_mask(const float* g, const void* mask, const float* y2, float* dz, float* partial, float* dzmax, int B, int T,
                            wm_stream_t stream);      /* dzmax (optional): B * 64 floats, max |dz| per row (wm_bn_bwd_finalize's gscale) */
/* A, Cc: [64]; Bc: [2][64] (hi, lo words of the offset, see wm_conv64 pro 3) */
int wm_bn_bwd_finalize(const float* partials, int nparts, double count, const float* gamma, const float* save_mean,
                       const float* save_invstd, float* A, float* Bc, float* Cc, float* dgamma, float* dbeta,
                       int accumulate, int eval_mode, const float* dzmax, int nmax, float* gscale, wm_stream_t stream);
/* gscale (optional, 2 floats; needs dzmax [nmax] = max |dz| per producer block): {gs, 1 / gs}, gs = the power of two that brings
 * max |A| * max |dz| to 2^9 -- the scale under which wm_dwgrad64_bf (arith 1) splits the rebuilt gradient into two f16 pieces. */

/* ---- stem / heads: Conv1d(1,64,7,p=3) :134,:177 ; Conv1d(64,1,1) :146 ; Conv1d(64,1+bits,1) :180 (+permute :186) */
int wm_stem_fwd(const float* s, const float* w, const float* bias, float* y, int B, int T, wm_stream_t stream);
int wm_stem_bwd(const float* g, const float* s, const float* w, float* ds, float* partial, float* dw, float* db, int B,
                int T, int nds, int accumulate, wm_stream_t stream);   /* ds rows only for clips [0, nds); partial >= 512*512 floats */
int wm_head1_fwd(const float* x, const float* w, const float* bias, float* y, int B, int T, wm_stream_t stream);
int wm_head1_bwd(const float* g, const float* x, const float* w, float* dx, float* partial, float* dw, float* db, int B,
                 int T, int accumulate, wm_stream_t stream);
/* logits written as (B,T,NO) contiguous -- the layout Detector.forward's permuted view exposes */
int wm_headN_fwd(const float* x, const float* w, const float* bias, float* y, int B, int T, int NO, wm_stream_t stream);
int wm_headN_bwd(const float* g, const float* x, const float* w, float* dx, float* partial, float* dw, float* db, int B,
                 int T, int NO, int accumulate, wm_stream_t stream);

/* ---- nn.LSTM(64,64,batch_first=True) :138,:152-154 ---------------------------------------------------------- */
int wm_lstm_xproj(const float* x, const float* w_ih, const float* b_ih, const float* b_hh, float* xp, int B, int T,
                  wm_stream_t stream);
int wm_lstm_fwd(const float* xp, const float* w_hh, float* hout, float* gates, float* cst, int B, int T, wm_stream_t stream);
/* xproj + recurrence in one launch: the projection of the next 32 steps runs on the bf16 matrix cores (bf16x6 split,
 * fp32-grade) beside the recurrence and never touches HBM.  gates [B,T,256] / cst [B,T,64] (both or neither): saved
 * activations / cell states for wm_lstm_bwd, exactly as wm_lstm_fwd writes them.  T >= 8. */
int wm_lstm_fwd_fused(const float* x, const float* w_ih, const float* b_ih, const float* b_hh, const float* w_hh,
                      float* hout, float* gates, float* cst, int B, int T, wm_stream_t stream);
/* build of wm_lstm_fwd_fused (process-wide; default 1): 1 wave-specialised -- the recurrence on waves 0..3, the projection of the next
 * 32-step chunk on four helper waves that share its one barrier per step; 0 the projection inside the recurrence's own waves.
 * Bit-identical results. */
int wm_set_lstm_fwd_wave_specialised(int on, wm_stream_t stream);
int wm_lstm_bwd(float* gates, const float* cst, const float* dh_out, const float* w_hh, int B, int T, wm_stream_t stream);
int wm_lstm_dx(const float* da, const float* w_ih, float* dx, int B, int T, wm_stream_t stream);
/* arithmetic of wm_lstm_dx and wm_lstm_wgrad (process-wide): 1 bf16x6 split on the bf16 matrix cores (default,
 * fp32-grade) | 0 native fp32 MFMA */
int wm_set_lstm_dx_bf16x6(int on, wm_stream_t stream);
/* wm_lstm_bwd + wm_lstm_dx in one launch: dx = da W_ih is formed chunk by chunk on the bf16 matrix cores (bf16x6 split)
 * beside the recurrence, from LDS -- no second pass over the [B,T,256] da tensor.  gates: activations in, da out. */
int wm_lstm_bwd_fused(float* gates, const float* cst, const float* dh_out, const float* w_hh, const float* w_ih, float* dx,
                      int B, int T, wm_stream_t stream);
int wm_lstm_wgrad(const float* da, const float* x, const float* h, float* partial, float* dw_ih, float* dw_hh,
                  float* db_ih, float* db_hh, int B, int T, int accumulate, wm_stream_t stream);
/* wm_lstm_bwd + wm_lstm_wgrad in one launch (py/main16.py:141,153 under autograd): four more waves per workgroup form the clip's
 * dW_ih / dW_hh on the matrix cores (bf16x6) out of the 32-step chunk of da the recurrence has just finished, from an LDS image --
 * da is not read back from HBM for them; bias gradients from the recurrence lanes.  gates: saved activations in, da out (still
 * written: wm_lstm_dx reads it).  partial >= B * (256*128 + 256) floats (one slab per clip, reduced in fixed order); accumulate as
 * wm_lstm_wgrad.  T % 32 == 0 and T >= 64 (hipErrorInvalidValue otherwise: use the two separate entry points). */
int wm_lstm_bwd_wgrad(float* gates, const float* cst, const float* dh_out, const float* w_hh, const float* x, const float* h,
                      float* partial, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, int B, int T, int accumulate,
                      wm_stream_t stream);

/* ---- nn.Embedding(2**bits,64) lookup :158 and its dense gradient ------------------------------------------- */
int wm_embed_gather(const float* table, const long long* message, float* vec, int B, int nrows, int* err, wm_stream_t stream);
int wm_embed_scatter_add(float* dtable, const long long* message, const float* dvec, int B, int nrows, wm_stream_t stream);
int wm_rowsum(const float* x, float* out, int rows, int T, wm_stream_t stream);

/* ---- delta post-processing: fir_lowpass :53-64 (bit 0), clamp_peak :66-67 (bit 1), limit_rms :69-72 (bit 2) -- */
int wm_postproc_fwd(const float* d_in, const float* taps, int ntaps, float thr, float max_rms, float eps, int stages,
                    float* f_out, float* d_out, float* stats, int B, int T, wm_stream_t stream);
int wm_postproc_bwd(const float* g, const float* f_in, const float* stats, const float* taps, int ntaps, float thr,
                    float max_rms, int stages, float* d_raw, int B, int T, wm_stream_t stream);

/* ---- STFT loss stack: MultiScaleMelLoss :192-202, TFLoudnessLoss :204-217, high_freq_penalty :74-81 ---------
 * loss_out: device scalar.  dsig (NULL = forward only): d loss / d (second signal | delta), [B,T].
 * gframes: scratch [B, 1+T/hop, n_fft]; partial: scratch [B*(1+T/hop)].                                          */
int wm_mel_loss(const float* clean, const float* wm, const float* fb, const int* klo, const int* khi, const int* mlo,
                float* gframes, float* partial, float* loss_out, float* dsig, int B, int T, wm_stream_t stream);
int wm_loud_loss(const float* clean, const float* wm, float thresh, float* gframes, float* partial, float* loss_out,
                 float* dsig, int B, int T, wm_stream_t stream);
int wm_hf_penalty(const float* delta, int kcut, float* gframes, float* partial, float* loss_out, float* dsig, int B, int T,
                  wm_stream_t stream);

/* ---- point-wise losses :252-266 and the optimizer update :504,:278 ------------------------------------------
 * wm_bce_*: logits [R = 2B][T][NO]; partial: >= 2 * R * ceil(T*NO / 4096) floats of scratch; T*NO < 2^23, R <= 65535 */
int wm_bce_fwd(const float* logits, const long long* message, float* partial, float* loc_out, float* bce_out, int B, int R,
               int T, int NO, wm_stream_t stream);
int wm_bce_bwd(const float* logits, const long long* message, const float* g_loc, const float* g_bce, float* dlogits, int B,
               int R, int T, int NO, wm_stream_t stream);
int wm_l1_fwd(const float* x, float* partial, float* out, long long n, wm_stream_t stream);
int wm_l1_bwd(const float* x, const float* g, float* dx, long long n, wm_stream_t stream);
/* step >= 1 is the 1-based update count; bias corrections are formed in double (torch.optim.Adam semantics) */
int wm_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2, double eps,
                 int step, wm_stream_t stream);

/* ---- main14b_2 deep-residual variant (py/main14b_2.py:83-224, BASELINE config 5): generic-shape convolutions -------
 * y[nb][co][t'] = act(bias[co] + vec[nb][co] + res[..] + sum_{ci,k} wp[ci*K+k][m] * x[nb][ci][n*S + k - P])
 *   st == 1: m = co, t' = n  (strided Conv1d :87-92, nn.Linear :134, Conv1d k7 :121,:139,:153)
 *   st  > 1: m = co*st + phase, t' = n*st + phase - shp  (ConvTranspose1d(k=2*st, stride st, padding st/2) :147 as a
 *            2-tap convolution + pixel shuffle).  act: 0 none | 1 ELU (:90) | 2 multiply by ELU'(y) with y read from `res` (the
 *            data gradient of the convolution behind an ELU, py/main14b_2.py:96-97, leaves the kernel as dL/dz).  wp is packed by the host mirror. */
/* x2 / Cin1 (optional, NULL / 0): input channels >= Cin1 are the rows of a second tensor x2 [NB][Cin - Cin1][Lin] -- two
 * gradients contracted by one launch (a strided Conv1d and its 1x1 skip conv into the same dL/dx).  nph (0 = st): phases per
 * output channel when st > 1 (m = co*nph + phase, phase < nph <= st): a strided convolution's data gradient only has K phases. */
int wm_gconv(const float* x, const float* wp, const float* bias, const float* vec, const float* res, float* y, int NB,
             int Cin, int Lin, int K, int S, int P, int Mtot, int Nout, int st, int shp, int Cout, int Lout, int act,
             const float* x2, int Cin1, int nph, wm_stream_t stream);
/* wm_gconv on the f16 two-piece split (three f16 piece products per product on v_mfma_f32_32x32x16_f16, fp32 accumulate, fp32-grade):
 * same arguments with wph = wm_gconv_pack_h's image of the SAME wp instead of wp; needs Cin % 16 == 0 (and Cin1 % 16 == 0 with a second
 * source).  gscale = {gs, 1 / gs} (wm_gscale_absmax) when x is a gradient -- it is multiplied by gs before the split and clamped to
 * +-6e4 --, NULL for activations (split unscaled). */
int wm_gconv_h(const float* x, const void* wph, const float* bias, const float* vec, const float* res, float* y, int NB,
               int Cin, int Lin, int K, int S, int P, int Mtot, int Nout, int st, int shp, int Cout, int Lout, int act,
               const float* x2, int Cin1, int nph, const float* gscale, float* ymax, wm_stream_t stream);
/* ymax (optional, ONE float zeroed by the caller): receives max |y| over everything the launch stores (atomic max) -- the gradient scale
 * of whatever consumes y next comes from wm_gscale_from_max(ymax, 1, ...) instead of a pass over y */
/* wph: 2 * Cin * K * Mtot f16 ([piece][Cin / 16][K][Mtot][16]: w * ws) followed by {ws, 1 / ws} as two floats.  scratch (>= 1024 floats):
 * ws = the power of two with max |w| ws in (2^9, 2^10] (a pass over wp); scratch NULL: ws = 2^8 fixed, one launch -- what the host mirror
 * uses: the scale only has to keep the two pieces inside the f16 range, which 2^8 does for max |w| between 4e-6 and 250 */
int wm_gconv_pack_h(const float* wp, void* wph, float* scratch, int Cin, int K, int Mtot, wm_stream_t stream);
/* the same image straight from a Conv1d weight w [Cout][Cin][K], fixed scale 2^8: mode 0 = the forward matrix (wp[ci K + k][co] = w[co][ci][k]),
 * mode 1 = the stride-1 data-gradient matrix (wp[co K + kk][ci] = w[co][ci][K - 1 - kk]); the GEMM channel count (Cin | Cout) % 16 == 0 */
int wm_gconv_pack_h_conv(const float* w, void* wph, int Cout, int Cin, int K, int mode, wm_stream_t stream);
/* generic weight gradient, one stride-1 GEMM with the taps folded into the column index (deterministic: split-K partial
 * tiles in `slab`, then a fixed-order fp64 reduce -- no float atomics):
 *   G[a][b][k] (+)= sum_{nb,t} A[nb][a][t] * Bx[nb][b][t + k - P],  dbias[a] (+)= sum A   (Conv1d stride 1: A = dL/dy,
 *   Bx = input; Linear / LSTM: K = 1).  Strided Conv1d / ConvTranspose1d: re-lay the strided operand with wm_gather_taps
 *   first (K = 1 / K = 2 problems) and let `remap` restore the weight's own order: 0 identity | 1 columns k*r1 + b ->
 *   b*r2 + k (r1 = Cin, r2 = taps) | 2 columns (co*r1 + ph)*2 + q -> co*2*r1 + q*r1 + ph (r1 = stride).
 *   b_clip_stride: floats between clips of Bx (0 = dense), so Bx may be a channel slice.  accumulate: 0 overwrite | 1 add.
 * wm_gwgrad_plan reports the workspace (fp32 elements) `slab` must hold for a problem shape.                            */
int wm_gwgrad_plan(int NB, int Ca, int Cb, int La, int K, long long* slab_floats, wm_stream_t stream);   /* host-only query; stream unused */
int wm_gwgrad(const float* A, const float* Bx, float* G, float* dbias, float* slab, int NB, int Ca, int Cb, int La, int Lb,
              int K, int P, long long b_clip_stride, int remap, int r1, int r2, int accumulate, wm_stream_t stream);
/* y[nb][row][t] = x[nb][c][t*S + k - P] (0 outside the clip), t < Lout, c < C, k < K; order 0: row = k*C + c (tap planes of
 * a strided Conv1d, py/main14b_2.py:87-92) | 1: row = c*K + k (stride phases of a ConvTranspose1d, :147)                */
int wm_gather_taps(const float* x, float* y, int NB, int C, int Lin, int K, int S, int P, int Lout, int order, wm_stream_t stream);
/* dz = g * ELU'(z) from y = ELU(z) (py/main14b_2.py:90,:96,:101) */
int wm_elu_bwd(const float* g, const float* y, float* dz, long long n, wm_stream_t stream);
/* out[c] (+)= sum_{nb,t} x[nb][c][t] in a fixed order (partial: >= 64*C floats of scratch; accumulate 0 | 1);
 * out[row] = sum_t x[row][t] for any row length */
int wm_channel_sum(const float* x, float* out, float* partial, int NB, int C, int L, int accumulate, wm_stream_t stream);
int wm_rowsum_any(const float* x, float* out, int rows, int L, wm_stream_t stream);
/* dense embedding gradient for any width: dtable[idx[b]][:] += dvec[b][:], duplicate ids added in batch order */
int wm_rows_scatter_add(float* dtable, const long long* idx, const float* dvec, int Bn, int dim, int nrows, wm_stream_t stream);
/* [A][C][L] -> [L][C][A]: batch-major <-> time-major sequence layout around nn.LSTM (:137) */
int wm_permute_acl(const float* x, float* y, int A, int C, int L, wm_stream_t stream);
/* one layer of nn.LSTM(hd, hd, num_layers=2) (py/main14b_2.py:137, :165) over all T steps as a chain of per-step launches
 * issued by the launcher (the kernel boundary is the step barrier): gate GEMM on the fp32 matrix cores + cell update.
 * Time-major, batch contiguous.  xp [T][4H][B] = W_ih x_t + b (activations overwrite it when save != 0); whh [4H][H];
 * hs, cs [T+1][H][B], row 0 = zero initial state (caller), h_t = hs[t+1].  H % 32 == 0, H <= 256.                   */
int wm_lstm_seq_fwd(float* xp, const float* whh, float* hs, float* cs, int T, int H, int B, int save, wm_stream_t stream);
/* BPTT of that layer: gates [T][4H][B] activations in -> pre-activation gradients out (in place); dout [T][H][B];
 * whhT = weight_hh^T [H][4H]; dc [H][B] scratch.                                                                     */
int wm_lstm_seq_bwd(float* gates, const float* cs, const float* dout, const float* whhT, float* dc, int T, int H, int B,
                    wm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* WM_HIP_H */
