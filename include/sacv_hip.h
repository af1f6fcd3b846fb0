/* libsacv_hip.so -- C ABI of the MI355X (gfx950) detection training step for SimpleAICV.
 *
 * The reference (zgcr/SimpleAICV-pytorch-ImageNet-COCO-training) has no FFI of its own on this
 * path: its seam is the Python class contract of simpleAICV/detection/{models,losses,decode}.
 * This ABI sits UNDER that contract.  Each entry point names the reference code it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked "host";
 *     the caller owns all buffers including workspaces (sizes from the *_workspace_* queries);
 *   - activations are NHWC fp32 ([N][H][W][C], C contiguous); conv weights are OIHW fp32
 *     (reference state_dict layout) and are repacked by sacv_pack_weight;
 *   - `stream` is a hipStream_t (0 = default stream); calls only enqueue work, never synchronise;
 *   - return 0 on success, negative on error (-1 bad argument, -2 launch failure,
 *     -3 unsupported); sacv_last_error() returns the message for the calling thread;
 *   - the shared object exports exactly the functions declared here (-fvisibility=hidden + csrc/exports.map;
 *     tests/test_abi.py compares the dynamic symbol table with this file).  sacv_abi_version() == 5.
 *
 * Where the symbol list of SURVEY.md 8(b) went (it was written before the kernels; arguments are positional scalars
 * and pointers here, not `const sacv_<op>_args*` records -- every call site is a ctypes call, a record would be packed
 * and unpacked for nothing):
 *   conv2d_fwd / conv2d_dgrad        sacv_conv2d_gemm (mode 0 / 1; grouped over pyramid levels), sacv_wino_gemm for the
 *                                    stride-1 3x3 head convolutions, sacv_stem_conv_fwd for the 7x7 stem
 *   conv2d_wgrad                     sacv_conv2d_wgrad, sacv_wino_wgrad, sacv_stem_conv_wgrad
 *   bn_train_fwd                     the statistics are an epilogue option of the producing convolution (SACV_CF_STATS)
 *                                    + sacv_bn_finalize_stats + sacv_bn_apply (sacv_bn_train_stats for a bare tensor)
 *   bn_train_bwd                     sacv_bn_train_bwd; its first stage can run in the epilogue of the data gradient that
 *                                    produces the upstream gradient (sacv_conv2d_gemm_bnbwd + sacv_bn_train_bwd_partial)
 *   sigmoid_permute_fwd / _bwd       no kernels: activations are NHWC, so the reference's permute(0,2,3,1).contiguous()
 *                                    is a view; sigmoid is SACV_CF_SIGMOID of the output convolution, its backward one
 *                                    mode of sacv_eltwise
 *   focal_fwd / focal_bwd            ONE pass: sacv_focal_loss_fwd_bwd writes the loss terms and d loss / d prob
 *   box_loss_fwd / box_loss_bwd      likewise: sacv_retina_box_loss_fwd_bwd, sacv_fcos_reg_ctr_loss_fwd_bwd
 *   decode_topk_nms                  sacv_decode_topk_nms, sacv_decode_topk_soft_nms (score scan + selection + box decode + NMS, batched)
 *   adamw_step                       sacv_adamw_step_gated (both parameter groups, skip gate and clip factor on the device)
 *   allreduce_bucket                 sacv_allreduce_bucket (RCCL on the communicator's own stream, sacv_comm_*)
 */
#ifndef SACV_HIP_H
#define SACV_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

const char* sacv_last_error(void);
int sacv_abi_version(void);
/* Stream `waiter` waits for everything enqueued on stream `signaller` so far (both hipStream_t of the current device): the
 * executor's replacement for torch.cuda.Stream.wait_stream / Event.record + wait_event between its launch stream and its
 * weight-gradient stream (the role of DDP's reducer events in the reference, tools/utils.py:178-180).  The event it records
 * is created with hipEventDisableTiming | hipEventDisableSystemFence: no system-scope cache writeback / invalidation per
 * record (nothing ordered this way leaves the device). */
int sacv_stream_wait_stream(void* waiter, void* signaller);
/* The same ordering through a ticket in device memory: a one-lane kernel on `signaller` stores it (release, agent scope),
 * a one-lane kernel on `waiter` polls it -- no barrier packet that depends on another queue's signal.  The wait kernel is
 * enqueued after its signal kernel (no deadlock on a shared hardware queue) and gives up after 2 s, counted by
 * sacv_stream_ticket_timeouts() (synchronises; 0 in a healthy process).  At most 8 signalling streams per device. */
int sacv_stream_wait_stream_ticket(void* waiter, void* signaller);
long long sacv_stream_ticket_timeouts(void);

/* Launch plan of the process's subsequent launches (one process drives one GPU; the plan is not per thread: a framework
 * may run the backward pass on another thread than the forward pass).  The convolution kernels are PERSISTENT: a launch is
 * sized to the workgroup slots it expects to own (slots per CU x CUs) and its tiles / stream-K ranges are cut for exactly
 * that many co-resident workgroups.  A caller that runs a second queue beside the launch stream -- the weight gradients on
 * a side HIP stream (the reference gets the same overlap from cuDNN + NCCL kernels co-running, tools/utils.py:178-180),
 * RCCL's all-reduce kernels -- states here what the launch stream may plan for, so that both queues' workgroups FIT a CU's
 * 160 KB of LDS and 512 registers per lane together instead of queueing behind each other:
 *   nt64 / nt128     slots per CU of sacv_conv2d_gemm's 128x64 / 128x128 tile kernels (alone: 3 / 2)
 *   wino             slots per CU of sacv_wino_gemm (alone: 3 narrow-item, 2 wide-item workgroups)
 *   wgrad            slots per CU the split planner of sacv_wino_wgrad fills (alone: 2); sacv_conv2d_wgrad takes its
 *                    split size (`chunk`) from the caller
 *   cu_reserve       CUs every persistent grid leaves free (RCCL's channels)
 *   split_policy     remainder tiles of sacv_conv2d_gemm (those left after the last full round of the persistent grid):
 *                    3 = cost model (one workgroup per tile / uniform K split / stream-K), 0 = never split along K
 *   wino_big         item shape of sacv_wino_gemm with split operands: 1 = cost model (whole rounds of 128 x 128 items,
 *                    one per CU, then 64 x 64 items for the remaining rows), 0 = 64 x 64 items only, 2 = every whole
 *                    128-row tile as 128 x 128 items
 *   nt_split         sacv_conv2d_gemm on the bf16 matrix pipe: the gathered fp32 operands are split into three bf16
 *                    terms (x = hi + mid + lo exactly) when they are stored to LDS and every product is formed from the
 *                    six partial products of weight >= 2^-16 in fp32 accumulators -- fp32 results (what is dropped is
 *                    below the rounding of one fp32 addition) at 6 x 32 instead of 8 x 64 matrix-pipe cycles per 16 K
 * A value <= 0 (cu_reserve, split_policy, wino_big, nt_split: < 0) keeps the current setting.  Results of a given plan are deterministic
 * (fixed-order sums); different plans differ by fp32 summation order. */
int sacv_launch_share(int nt64_wg_per_cu, int nt128_wg_per_cu, int wino_wg_per_cu, int wgrad_wg_per_cu, int cu_reserve,
                      int split_policy, int wino_big, int nt_split);

/* ---- convolution (nn.Conv2d: classification/backbones/resnet.py:33-39; detection/models/fpn.py:20-58;
 *      detection/models/head.py:20-33,60-73) -------------------------------------------------------- */

/* epilogue flags */
#define SACV_CF_BIAS 1     /* + bias[n]                                   */
#define SACV_CF_RELU 2     /* nn.ReLU fused (head.py:27,67)               */
#define SACV_CF_SIGMOID 4  /* nn.Sigmoid fused (head.py:50)               */
#define SACV_CF_ACCUM 8    /* dst += result (gradient fan-in)             */
#define SACV_CF_MASK 16    /* result *= (mask > 0)  (ReLU backward fused) */
#define SACV_CF_STATS 32   /* per-M-tile column sum / sum of squares of the stored values -> stats
                              (BatchNorm batch statistics of y; bias gradient when the output is a dy) */
#define SACV_CF_AFFINE 128 /* with SACV_CF_BIAS: bias is [2][N], result = acc * bias[1][n] + bias[0][n]: an eval-mode
                            BatchNorm (scale, shift from sacv_bn_fold_batched) folded into the convolution */

/* Grouped implicit-GEMM convolution on the fp32 matrix cores.  mode 0 = forward
 * (src = x, w = pack kind 0), mode 1 = data gradient (src = dy, w = pack kind 1).
 * Up to 8 problems that share the weights (the 5 FPN levels of one head layer) run in one launch.
 * Problem i: src[i] NHWC [Nb[i]][Hs[i]][Ws[i]][Cs]; dst[i] [Nb[i]*Hd[i]*Wd[i]][N].
 * remap_s > 0 (single problem): GEMM row (n,h,w) is written to row (n, h*remap_s, w*remap_s) of an
 * [Nb][Hr][Wr][N] tensor -- the data gradient of a strided 1x1 convolution. */
int sacv_conv2d_gemm(int nprob, const float* const* src /*host array*/, float* const* dst /*host array*/,
                     const float* const* mask /*host array or NULL*/, const int* Nb, const int* Hs, const int* Ws,
                     const int* Hd, const int* Wd /*host arrays*/, const float* w, const float* bias, int Cs, int N,
                     int KH, int KW, int stride, int pad, int mode, int flags, int remap_s, int Hr, int Wr,
                     const int* ext /*host int[12] or NULL, see below*/, float* workspace /*sacv_conv2d_gemm_workspace_bytes() or NULL*/,
                     float* stats /*[sacv_conv2d_gemm_stats_rows()][2][N], SACV_CF_STATS only*/, void* stream);
/* The kernel is persistent (2 workgroups per CU): whole output tiles are dealt round-robin and the
 * tiles left over after the last full round are split along K over the idle workgroups; the K-slices
 * meet in `workspace` and are summed in a fixed order (deterministic). */
/* ext (optional) = {pad_h, pad_w, kh0, khs, kw0, kws, KHreal, KWreal, 0, remap_oh, remap_ow, 0}: the KH x KW
 * tap loop reads weight taps (kh0 + khs*kh, kw0 + kws*kw) of a KHreal x KWreal panel, with separate pads and
 * a row-scatter offset.  Used for the data gradient of a stride-2 3x3 convolution, which is run as four
 * stride-1 problems, one per output-pixel parity class, each touching only the taps that reach it (2.25 taps
 * per pixel on average instead of 9). */
long long sacv_conv2d_gemm_workspace_bytes(void);
int sacv_conv2d_gemm_stats_rows(int nprob, const int* Nb, const int* Hd, const int* Wd);
/* One problem of sacv_conv2d_gemm -- a data gradient (mode 1) in the backward pass of conv -> BatchNorm -> ReLU
 * (classification/backbones/resnet.py:33-48; autograd's native_batch_norm_backward in the reference) -- whose output
 * dz [M][N] is the upstream gradient of a BatchNorm layer with input bn_y [M][N]: the epilogue also writes that layer's
 * backward partial sums, stats[m tile] = (sum g, sum g * xhat) with g = dz * (z > 0) and xhat = (y - mean) * invstd, for
 * sacv_bn_train_bwd_partial.  Mask: bn_z (the stored layer output), or recomputed from y with bn_gamma / bn_beta when bn_z
 * is NULL (bn_beta NULL too: no ReLU).  flags: SACV_CF_ACCUM / SACV_CF_MASK / SACV_CF_BIAS; stats:
 * sacv_conv2d_gemm_stats_rows(1, ...) rows of [2][N]. */
int sacv_conv2d_gemm_bnbwd(const float* src, float* dst, const float* mask, int Nb, int Hs, int Ws, int Hd, int Wd,
                           const float* w, const float* bias, int Cs, int N, int KH, int KW, int stride, int pad, int mode,
                           int flags, float* workspace, float* stats, const float* bn_y, const float* bn_z,
                           const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                           void* stream);

/* ---- Winograd F(2x2,3x3) path of the stride-1 3x3 head convolutions (head.py:20-36, 55-70, 103-141: nn.Conv2d
 *      3x3 / pad 1 over the 5 pyramid levels).  Three steps: weights -> U (once per optimizer step), input -> V,
 *      16 chained GEMMs + output transform + the sacv_conv2d_gemm epilogue in one kernel.  2.25x fewer
 *      matrix-core flops than sacv_conv2d_gemm; same results up to fp32 rounding.
 *      `split` (0 / 1) selects the operand form of U and V.  0: fp32 [16][rows][pad_k], multiplied on the fp32 matrix
 *      pipe.  1: three bf16 planes [16][3][rows][pad_k] with x = hi + mid + lo exactly (hi = bf16(x), mid = bf16(x - hi),
 *      lo = bf16(x - hi - mid)); the GEMM sums the six products of weight >= 2^-16 on the bf16 pipe (16x the fp32
 *      pipe's rate) into fp32 accumulators -- what it drops is below the rounding of one fp32 addition, the result has
 *      the fp32 kernel's error against fp64 (tests/test_gpu_winograd.py).  Operand bytes: 6 per element instead of 4.
 *      U, V and the `split` argument of the GEMM must agree. --------------------------------------------------- */
/* tiles of 2x2 output pixels over the pyramid: sum_l Nb * ceil(H[l]/2) * ceil(W[l]/2) */
int sacv_wino_tiles(int nlev, int Nb, const int* H, const int* W /*host arrays*/);
/* U[16][pad_n(N)][pad_k(K)] = G w G^T from a GEMM panel [N][9][K] (forward panel; data-gradient panel with flip = 1) */
int sacv_wino_weights(const float* panel, void* U, int N, int K, int flip, int split, void* stream);
/* njobs (<= 32) such transforms in one launch (all Winograd layers of a model, both directions, once per step) */
int sacv_wino_weights_batched(int njobs, const float* const* panel /*host arrays*/, void* const* U, const int* N,
                              const int* K, const int* flip, int split, void* stream);
/* V[16][Tpad][pad_k(C)] = B^T d B of every 4x4 input patch; Tpad = tiles rounded up to a multiple of 64 */
int sacv_wino_input(int nlev, const float* const* src /*host array*/, int Nb, const int* H, const int* W, int C,
                    void* V, int Tpad, int split, void* stream);
int sacv_wino_stats_rows(int tiles);
/* operand padding of U / V: K (input channels) to a multiple of 32, N (rows of U) to a multiple of 128, zero-filled */
int sacv_wino_pad_k(int K);
int sacv_wino_pad_n(int N);
/* nprob (1 or 2) independent convolutions over the same pyramid geometry (e.g. layer i of both head towers):
 * dst[p*5 + l] = [Nb][H[l]][W[l]][N[p]]; flags[p] = SACV_CF_* bits; mask[p*5 + l] with SACV_CF_MASK;
 * stats[p] = [sacv_wino_stats_rows][2][N[p]] with SACV_CF_STATS */
int sacv_wino_gemm(int nprob, int nlev, int Nb, const int* H, const int* W, const void* const* V /*host array*/,
                   const void* const* U /*host array*/, const int* Tpad, const int* K, const int* N,
                   const float* const* bias /*host array or NULL*/, float* const* dst /*host array [nprob*5]*/,
                   const float* const* mask /*host array [nprob*5] or NULL*/, float* const* stats /*host array or NULL*/,
                   const int* flags, int split, void* stream);

/* weight gradient on the same path: dw (+)= G^T [ sum_tiles (A dY A^T) (x) V ] G from V = sacv_wino_input(x) kept
 * from the forward pass and dy[l] = [Nb][H[l]][W[l]][Cout]; 16 split-K GEMMs in one launch + a fixed-order reduction
 * (deterministic).  dw_layout as sacv_conv2d_wgrad.  Cin (padded to 32) must be a multiple of 64.  v_split = 1: V is
 * the split form and the kernel rebuilds the fp32 values from the three planes, exactly: same dw bit for bit as with
 * v_split = 0.  v_split = 2: V is the split form and the GEMMs run on the bf16 pipe (A dY A^T is split the same way,
 * six products into fp32 accumulators; Cout % 8 == 0).  The workspace holds either form of the transformed dy. */
long long sacv_wino_wgrad_workspace_bytes(int Tpad, int Cout, int Cin);
int sacv_wino_wgrad(int nlev, const float* const* dy /*host array*/, int Nb, const int* H, const int* W, int Cout,
                    const void* V, int v_split, int Tpad, int Cin, float* dw, int accumulate, int dw_layout,
                    float* workspace, void* stream);

/* pixels per split (`chunk`) that fills the weight-gradient queue's workgroup slots for this problem list: npix[i] =
 * Nb * Ho * Wo of segment i, N2 = KH * KW * Cin_pad, taps = KH * KW */
int sacv_conv2d_wgrad_chunk(int nseg, const int* npix /*host*/, int Cout, int N2, int taps);
long long sacv_conv2d_wgrad_workspace_bytes(int nseg, const int* Nb, const int* Ho, const int* Wo, int Cin_pad,
                                            int Cout, int KH, int KW, int chunk);

/* Weight gradient dw[OIHW] (+)= sum over segments of x (*) dy, split over pixels in `chunk`-sized
 * slabs and reduced in a fixed order (deterministic).  dw_taps > 0: x is the stem im2col matrix
 * (KH = KW = 1, Cin_pad = padded K) and dw is [Cout][Cin][dw_taps].  dw_layout 1: dw is stored
 * [Cout][KH][KW][Cin] (channels_last weight gradient = the forward GEMM panel order) instead of OIHW. */
int sacv_conv2d_wgrad(int nseg, const float* const* dy, const float* const* x, const int* Nb, const int* Hx,
                      const int* Wx, const int* Ho, const int* Wo /*host arrays*/, float* dw, float* workspace,
                      int Cin, int Cin_pad, int Cout, int KH, int KW, int stride, int pad, int chunk, int accumulate,
                      int dw_taps, int dw_layout, void* stream);

/* kind 0: [Cout][KH*KW*Cin] forward panel; 1: [Cin][KH*KW*Cout] data-gradient panel;
 * 2: stem panel [Cout][Kpad] (k = tap*Cin + ci, zero padded); 3: as 1 from a channels_last ([Cout][KH][KW][Cin])
 * source.  A channels_last weight IS its forward panel: no pack is needed for kind 0 then. */
int sacv_pack_weight(const float* w_oihw, float* out, int Cout, int Cin, int KH, int KW, int kind, int Kpad,
                     void* stream);
/* every data-gradient panel of a model in one launch.  descs: device array of 32-byte records
 * {const float* src; float* dst; int Cout, Cin, T, src_ohwi}; chunk_map: device int[nchunks][2] =
 * (record index, 32x32 tile index = (tap * ceil(Cin/32) + ci_block) * ceil(Cout/32) + co_block). */
int sacv_pack_dgrad_batched(const void* descs, const int* chunk_map, int nchunks, void* stream);

/* im2col of the 7x7/s2 stem (backbones/resnet.py:42-49).  x is addressed by element strides, so both
 * NCHW-contiguous and the collater's channels_last images (detection/common.py:262-268) are accepted. */
int sacv_im2col_stem(const float* x, float* col, int Nb, int Cin, int H, int W, long long sN, long long sC,
                     long long sH, long long sW, int Ho, int Wo, int KH, int KW, int stride, int pad, int Kpad,
                     void* stream);
/* The 7x7/s2/p3 stem convolution itself (classification/backbones/resnet.py:176-181, the `conv1` of
 * detection/models/backbones/resnet.py:96-100), forward, without the im2col matrix: x as above (3 channels),
 * wpanel = the packed stem panel [64][Kpad] (sacv_pack_weight stem mode, k = (ky*7+kx)*3 + c), y [Nb][Ho][Wo][64];
 * stats (optional) [sacv_stem_conv_stats_rows][2][64] = per-tile column sums / sums of squares for BatchNorm. */
int sacv_stem_conv_stats_rows(int Nb, int Ho, int Wo);
int sacv_stem_conv_fwd(const float* x, int Nb, int H, int W, long long sN, long long sC, long long sH, long long sW,
                       const float* wpanel, int Kpad, float* y, int Ho, int Wo, float* stats, void* stream);
/* ... and its weight gradient, from the image and dy [Nb][Ho][Wo][64] (the backward of the same nn.Conv2d; autograd's
 * convolution_backward in the reference): dw [64][3][7][7] (OIHW) (+)= x (*) dy.  workspace: one [64][160] slab per
 * persistent workgroup, summed in a fixed order. */
long long sacv_stem_conv_wgrad_workspace_bytes(int Nb, int Ho, int Wo);
int sacv_stem_conv_wgrad(const float* x, int Nb, int H, int W, long long sN, long long sC, long long sH, long long sW,
                         const float* dy, int Ho, int Wo, float* dw, float* workspace, int accumulate, void* stream);

/* ---- BatchNorm2d (classification/backbones/resnet.py:41; training mode batch statistics) ------------ */
/* eval mode: (shift, scale) of every BatchNorm of a model in one launch, for SACV_CF_AFFINE.  jobs = device array of
 * {const float* gamma, beta, running_mean, running_var; float* out; int C; int pad} (48 bytes), out = [2][C] */
int sacv_bn_fold_batched(const void* jobs, int njobs, float eps, void* stream);
long long sacv_bn_workspace_floats(int R, int C);
int sacv_bn_train_stats(const float* y, int R, int C, float eps, float momentum, float* mean, float* invstd,
                        float* running_mean, float* running_var, long long* num_batches_tracked, float* workspace,
                        void* stream);
/* second stage only, from the [nblocks][2][C] partials of a SACV_CF_STATS convolution */
int sacv_bn_finalize_stats(const float* partial, int nblocks, int R, int C, float eps, float momentum, float* mean,
                           float* invstd, float* running_mean, float* running_var, long long* num_batches_tracked,
                           void* stream);
/* z = act((y-mean)*invstd*gamma + beta [+ residual]); stat_is_var=1: eval mode with running_var. */
int sacv_bn_apply(const float* y, const float* mean, const float* invstd_or_var, const float* gamma,
                  const float* beta, const float* residual, float* z, long long numel, int C, int relu,
                  int stat_is_var, float eps, void* stream);
/* zmask != NULL fuses the ReLU backward (resnet.py:42, :153); with zmask == NULL and beta != NULL the ReLU mask
 * is recomputed from y (output = relu(bn(y)), no residual), which saves reading z; g_out receives the masked
 * upstream gradient (the residual branch's gradient). */
int sacv_bn_train_bwd(const float* dz, const float* zmask, const float* y, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, float* dy, float* g_out, float* dgamma, float* dbeta,
                      int R, int C, int accumulate, float* workspace, void* stream);
/* The same with its first stage (the per-channel sums of g and g * xhat over the rows) done elsewhere -- by the
 * epilogue of the data gradient that PRODUCES dz (sacv_conv2d_gemm_bnbwd): partial = [nparts][2][C]. */
int sacv_bn_train_bwd_partial(const float* dz, const float* zmask, const float* y, const float* mean,
                              const float* invstd, const float* gamma, const float* beta, float* dy, float* g_out,
                              float* dgamma, float* dbeta, int R, int C, int accumulate, const float* partial, int nparts,
                              float* workspace, void* stream);

/* ---- GroupNorm(32, C) + ReLU of the FCOS head towers (detection/models/head.py:110,125) --------------- */
long long sacv_gn_workspace_floats(int N, int HW, int C);
/* x [N][HW][C]; stats out [N][G][2] = (mean, invstd); z = relu?(groupnorm(x)) */
int sacv_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* z, float* stats, int N, int HW,
                       int C, int G, float eps, int relu, float* workspace, void* stream);
int sacv_groupnorm_bwd(const float* x, const float* dz, const float* zmask, const float* stats, const float* gamma,
                       float* dx, float* dgamma, float* dbeta, int N, int HW, int C, int G, int accumulate,
                       float* workspace, void* stream);

/* ---- pooling / resize / elementwise ------------------------------------------------------------------ */
/* nn.MaxPool2d(3,2,1) (detection/models/backbones/resnet.py:50); idx = window position of the max */
int sacv_maxpool_fwd(const float* x, float* y, unsigned char* idx, int Nb, int H, int W, int C, int Ho, int Wo,
                     int k, int s, int p, void* stream);
int sacv_maxpool_bwd(const float* dy, const unsigned char* idx, float* dx, int Nb, int H, int W, int C, int Ho,
                     int Wo, int k, int s, int p, void* stream);
/* F.interpolate(bilinear, align_corners=True) + lateral (fpn.py:65-73) */
int sacv_upsample_bilinear_add_fwd(const float* top, const float* lateral, float* out, int Nb, int Hs, int Ws,
                                   int Hd, int Wd, int C, void* stream);
int sacv_upsample_bilinear_bwd(const float* dout, float* dtop, int Nb, int Hs, int Ws, int Hd, int Wd, int C,
                               int accumulate, void* stream);
/* op 0 relu(a); 1 a*(b>0); 2 a*b*(1-b); 3 a+b; 4 a*scalar; 5 a*scalar*b*(1-b) */
int sacv_eltwise(const float* a, const float* b, float* out, long long n, int op, float scalar, void* stream);
long long sacv_colsum_workspace_floats(int R, int C);
int sacv_colsum(const float* x, float* out, int R, int C, int accumulate, float* workspace, void* stream);
int sacv_colsum_finalize(const float* partial, int nblocks, int stride, int C, float* out, int accumulate,
                         void* stream);
/* n (<= 16) such second stages in one launch (the parameter gradients of one transformer block) */
int sacv_colsum_finalize_multi(int n, const float* const* partial /*host arrays*/, const int* nblocks, const int* stride,
                               const int* C, float* const* out, const int* accumulate, void* stream);

/* ---- targets and losses (detection/models/anchor.py, detection/losses.py) ---------------------------- */
/* RetinaAnchors.__call__ (anchor.py:18-86): bit-exact vs NumPy fp32; fw/fh/areas/... are host arrays */
int sacv_retina_anchors(int nlev, const int* fw, const int* fh, const float* areas, const float* ratios,
                        int nratios, const float* scales, int nscales, const float* strides, float* anchors_out,
                        void* stream);
/* FCOSPositions.__call__ (anchor.py:94-130) */
int sacv_fcos_points(int nlev, const int* fw, const int* fh, const float* strides, float* points_out,
                     void* stream);
/* RetinaLoss.get_batch_anchors_annotations (losses.py:322-409): targets [B][A][5], counts int[2] */
int sacv_retina_assign(const float* anchors, const float* annots, float* targets, int* counts, int B, int A, int G,
                       int smooth_l1, void* stream);
long long sacv_loss_workspace_floats(void);
/* compute_batch_focal_loss (losses.py:220-261 / :513-548) forward + gradient in one pass */
int sacv_focal_loss_fwd_bwd(int nlev, const float* const* cls_in, float* const* grad, const int* level_off,
                            const float* targets, int tstride, int B, int A, int C, float alpha, float gamma,
                            float weight, const int* counts, float* workspace, float* loss_out, int slot,
                            void* stream);
/* compute_batch_box_loss (losses.py:263-320): box_type 0 SmoothL1, 1 IoU, 2 GIoU, 3 DIoU, 4 CIoU, 5 EIoU
 * (IoUMethod, losses.py:28-123) */
int sacv_retina_box_loss_fwd_bwd(int nlev, const float* const* reg_in, float* const* grad, const int* level_off,
                                 const float* targets, const float* anchors, int B, int A, int box_type, float beta,
                                 float weight, const int* counts, float* workspace, float* loss_out, int slot,
                                 void* stream);

/* FCOSLoss.get_batch_position_annotations (losses.py:612-833): targets [B][P][8] =
 * (l,t,r,b, cls+1 | 0, centerness, px, py); level_off/strides/mi are host arrays; counts int[2] */
int sacv_fcos_assign(int nlev, const int* level_off, const float* strides, const float* mi, const float* points,
                     const float* annots, float* targets, int* counts, int B, int P, int G,
                     float center_sample_radius, int use_center_sample, void* stream);
/* compute_batch_iou_loss (losses.py:550-586, iou_type 0 IoU / 1 GIoU / 2 DIoU / 3 CIoU / 4 EIoU, weighted by the
 * centerness target) and
 * compute_batch_centerness_loss (:588-610), forward + gradients */
int sacv_fcos_reg_ctr_loss_fwd_bwd(int nlev, const float* const* reg_in, const float* const* ctr_in,
                                   float* const* dreg, float* const* dctr, const int* level_off,
                                   const float* targets, int B, int P, int iou_type, float w_reg, float w_ctr,
                                   const int* counts, float* workspace, float* loss_out, int slot_reg, int slot_ctr,
                                   void* stream);
/* FCOS output split (fcos.py:70-80; head.py:176-179): y8 [R][8] = (reg_raw[4], centerness logit, pad) ->
 * reg = reg_raw * exp(scale), ctr = sigmoid(logit); and its backward (+ d loss / d scale) */
int sacv_fcos_split_fwd(const float* y8, const float* scale, float* reg, float* ctr, int R, void* stream);
int sacv_fcos_split_bwd(const float* dreg, const float* dctr, const float* reg, const float* ctr, const float* scale,
                        float gscale, float* dy8, float* dscale, int idx, int accumulate, int R, float* workspace,
                        void* stream);

/* ---- postprocess (detection/decode.py) -------------------------------------------------------------- */
long long sacv_decode_workspace_bytes(int B, int A, int topn);
/* RetinaDecoder / FCOSDecoder + DecodeMethod + DetNMSMethod (decode.py:26-364).
 * kind 0 RetinaNet (geom = anchors), 1 FCOS (geom = points); nms_type 0 python_nms, 1 diou_python_nms,
 * 2 torch_nms (torchvision.ops.nms as documented: unclamped IoU, drop when IoU > threshold) */
int sacv_decode_topk_nms(int kind, int nlev, const float* const* cls, const float* const* reg,
                         const float* const* ctr, const int* level_off, const float* geom, int B, int A, int C,
                         float min_score, int topn, int nms_type, float nms_thr, int max_obj, void* workspace,
                         float* out_scores, float* out_classes, float* out_boxes, void* stream);
/* The same front end followed by batched Soft-NMS (Bodla et al., ICCV 2017) instead of the hard suppression of
 * DetNMSMethod (decode.py:26-104).  The reference has NO soft-NMS: this is the extra mode BASELINE.json's north_star
 * names, an extension of DecodeMethod's nms_type (decode.py:121-172), parity unpinned.  method 0 linear (decay by
 * 1 - IoU when IoU >= nms_thr), 1 gaussian (exp(-IoU^2 / sigma)); a box leaves the candidate set when its decayed
 * score is <= prune_score; IoU as python_nms (decode.py:45-72).  Same workspace, same outputs (scores = decayed). */
int sacv_decode_topk_soft_nms(int kind, int nlev, const float* const* cls, const float* const* reg,
                              const float* const* ctr, const int* level_off, const float* geom, int B, int A, int C,
                              float min_score, int topn, int method, float nms_thr, float sigma, float prune_score,
                              int max_obj, void* workspace, float* out_scores, float* out_classes, float* out_boxes,
                              void* stream);

/* ---- data boundary either side of the step ------------------------------------- */
/* DetectionCollater + Normalize (detection/common.py:223-286): B (<=16) uint8 [h][w][3] device images ->
 * zero-padded fp32 batch [B][R][R][3] = channels_last storage of the model input, value / 255 */
int sacv_collate_u8(int B, const unsigned char* const* imgs /*host array*/, const int* h, const int* w, int R,
                    float* out, void* stream);
/* the same on a rectangular canvas [B][Hc][Wc][3] (Hc x Wc >= every image): the opt-in padding-skip form of
 * DetectionCollater (detection/common.py:251-268 pads every batch to R x R) */
int sacv_collate_u8_rect(int B, const unsigned char* const* imgs /*host array*/, const int* h, const int* w, int Hc,
                         int Wc, float* out, void* stream);
/* evaluate_coco_detection post-steps (tools/scripts.py:742-758), in place on boxes [B][M][4]:
 * / scale, clip to the original image size, xyxy -> xywh */
int sacv_eval_postprocess(float* boxes, const float* scales, const float* sizes, int B, int M, void* stream);

/* ---- ResNet50 classification tail (classification/backbones/resnet.py:203,241;
 *      classification/losses.py:14-26).  The FC layer runs as a 1x1 problem of sacv_conv2d_gemm. ------------- */
int sacv_avgpool_fwd(const float* x, float* out, int N, int HW, int C, void* stream);
int sacv_avgpool_bwd(const float* dy, float* dx, int N, int HW, int C, void* stream);
/* nn.CrossEntropyLoss(reduction='mean') forward + gradient; labels int64; workspace >= ceil(N/4) floats */
int sacv_softmax_ce_fwd_bwd(const float* logits, const long long* labels, float* dlogits, float* loss_out, int N,
                            int C, float grad_scale, float* workspace, void* stream);
/* torch.optim.SGD with momentum over the flat arena ([0,split) uses wd_lo, the rest wd_hi) */
int sacv_sgd_step(float* params, const float* grads, float* momentum_buf, long long n, long long split, float lr,
                  float momentum, float wd_lo, float wd_hi, int first_step, float grad_scale, void* stream);

/* ---- optimizer (tools/utils.py:480-582 build_optimizer -> torch.optim.AdamW) ------------------------- */
int sacv_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                    long long split, float lr, float beta1, float beta2, float eps, float wd_lo, float wd_hi,
                    int step, float grad_scale, void* stream);
/* step gate of the training loop (tools/scripts.py:852-1050), evaluated on the device (no host round trip):
 *   sacv_fill(flag, 1, 0) -> sacv_nonfinite_flag(images / annotations: the inf/nan input guard, :878-884)
 *   -> sacv_loss_skip_flag(loss terms: zero / inf / nan loss, :918-930) [-> sacv_nonfinite_flag(grads, nan_only):
 *   skip_inf_nan_grad, :951-961] -> (all-reduce SUM of flag: a batch bad on any rank is skipped by all, :963-974)
 *   -> sacv_step_gate -> [sacv_grad_clip_coef: clip_grad_value_ / clip_grad_norm_, :978-990] -> *_step_gated.
 * state = float[8] {optimizer steps taken, skip this step, 1-beta1^t, sqrt(1-beta2^t), clip multiplier,
 *                   gradient norm, 0, 0}. */
int sacv_nonfinite_flag(const float* x, long long n, int nan_only, float* flag, void* stream);
int sacv_loss_skip_flag(const float* values, int n, float* flag, int accumulate, void* stream);
int sacv_step_gate(const float* flag, float* state, float beta1, float beta2, void* stream);
long long sacv_grad_clip_workspace_bytes(void);
int sacv_grad_clip_coef(const float* grads, long long n, float grad_scale, float clip_value, float max_norm,
                        float* state, void* workspace, void* stream);
/* [0,split) = the 1-D tensors (lr_lo, wd_lo), the rest the >=2-D tensors (lr_hi, wd_hi): the two param_groups of
 * build_optimizer (tools/utils.py:480-582), whose 'lr' entries Scheduler.step rewrites (tools/utils.py:199-223) */
int sacv_adamw_step_gated(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                          long long split, float lr_lo, float lr_hi, float beta1, float beta2, float eps, float wd_lo,
                          float wd_hi, const float* state, float grad_scale, float clip_value, void* stream);
int sacv_sgd_step_gated(float* params, const float* grads, float* momentum_buf, long long n, long long split,
                        float lr_lo, float lr_hi, float momentum, float wd_lo, float wd_hi, int nesterov,
                        const float* state, float grad_scale, float clip_value, void* stream);
/* EmaModel.update (tools/utils.py:150-163) over a flat block: ema = decay * ema + one_minus_decay * model in torch's
 * arithmetic (the caller passes float32(decay) and float32(1. - decay), two products and a sum, each rounded); state
 * (optional): the step state of sacv_step_gate -- nothing is written when the step was skipped */
int sacv_ema_update(float* ema, const float* model, long long n, float decay, float one_minus_decay, const float* state,
                    void* stream);
int sacv_fill(float* p, long long n, float value, void* stream);

/* ---- data-parallel exchange over RCCL / xGMI (replaces nn.parallel.DistributedDataParallel of
 *      tools/utils.py:166-185; process group of tools/train_detection_model.py:50-59).  One communicator per
 *      process = per GPU.  RCCL is dlopen'ed on first use (librccl.so.1, a copy already in the process first). ---- */
int sacv_comm_unique_id_bytes(void);
/* rank 0: rendezvous id into a HOST buffer of sacv_comm_unique_id_bytes() bytes; pass it to the other ranks */
int sacv_comm_unique_id(void* id_out /*host*/);
/* collective; binds to the calling thread's current HIP device */
int sacv_comm_init(const void* id /*host*/, int rank, int world, void** comm_out);
int sacv_comm_destroy(void* comm);
/* gradient bucket buf[0..count) <- sum over ranks, in place, on the communicator's side HIP stream, which first
 * waits (event) for what `producer_stream` holds so far: the transfer overlaps with later backward kernels */
int sacv_allreduce_bucket(void* comm, float* buf, long long count, void* producer_stream);
/* `consumer_stream` (the optimizer's) waits for every bucket enqueued so far; no host synchronisation */
int sacv_comm_wait(void* comm, void* consumer_stream);
/* Measurement aid: `workgroups` resident workgroups (256 threads, lds_bytes of LDS each) that stream `scratch` for `us`
 * microseconds on `stream` -- a stand-in for RCCL's kernels when the CU reserve (sacv_launch_share) is sized on a box
 * with one GPU. */
int sacv_comm_emulate_load(int workgroups, int lds_bytes, float us, float* scratch, long long scratch_bytes,
                           void* stream);
/* DDP init_sync / broadcast_buffers: rank `root`'s buf overwrites every rank's, in order on `stream` */
int sacv_comm_broadcast(void* comm, float* buf, long long count, int root, void* stream);

/* ---- bf16 matrix-core path (the autocast ResNet50 ImageNet recipe, tools/scripts.py:108-267 with
 *      use_amp, 0.classification_training/imagenet/resnet50/train_config.py:97).  Activations are NHWC bf16 in HBM,
 *      accumulation, BatchNorm statistics, losses, master weights, gradients and optimizer state are fp32.
 *      `const void*` activation / weight-panel pointers are bf16 device arrays. ------------------------------------ */
#define SACV_CFH_OUT_F32 64 /* sacv_conv2d_gemm_bf16: dst is fp32 (the logits) */
/* forward (mode 0) / data gradient (mode 1) of one convolution on v_mfma_f32_32x32x16_bf16; w = bf16 panel
 * [N][KH*KW*Cs] (sacv_cast_bf16 of a channels_last / 1x1 weight; sacv_pack_dgrad_batched_bf16 for mode 1);
 * flags SACV_CF_BIAS | SACV_CF_ACCUM | SACV_CF_STATS | SACV_CFH_OUT_F32; remap / ext as sacv_conv2d_gemm.
 * SACV_CF_STATS sums the bf16-ROUNDED outputs (what BatchNorm reads). */
int sacv_conv2d_gemm_bf16(const void* src, void* dst, int Nb, int Hs, int Ws, int Hd, int Wd, const void* w,
                          const float* bias, int Cs, int N, int KH, int KW, int stride, int pad, int mode, int flags,
                          int remap_s, int Hr, int Wr, const int* ext, float* stats, void* stream);
/* nn.Linear on bf16 rows (vit.py:139-160, vit_mae.py): y[M][N] (bf16, or fp32 with SACV_CFH_OUT_F32) =
 * x[M][K] w[N][K]^T (+ bias) (+ residual: an fp32 [M][N] tensor, the residual stream -- the block's `x + proj(...)`
 * without a copy of x) (+ y with SACV_CF_ACCUM).  Large problems run on the LDS-DMA pipelined GEMM (gemm_bf16.hip). */
int sacv_linear_bf16(const void* x, const void* w, void* y, const float* bias, const float* residual, long long M,
                     int N, int K, int flags, void* stream);
long long sacv_conv2d_wgrad_bf16_workspace_bytes(int Nb, int Ho, int Wo, int Cin_pad, int Cout, int KH, int KW,
                                                 int chunk);
/* fp32 weight gradient from bf16 dy / x (fp32 slabs per `chunk` pixels, a multiple of 64, summed in fixed order) */
int sacv_conv2d_wgrad_bf16(const void* dy, const void* x, int Nb, int Hx, int Wx, int Ho, int Wo, float* dw,
                           float* workspace, int Cin, int Cin_pad, int Cout, int KH, int KW, int stride, int pad,
                           int chunk, int accumulate, int dw_taps, int dw_layout, void* stream);
int sacv_cast_bf16(const float* in, void* out, long long n, void* stream); /* round to nearest even */
int sacv_cast_f32(const void* in, float* out, long long n, void* stream);
/* as sacv_pack_dgrad_batched with fp32 sources and bf16 destinations */
int sacv_pack_dgrad_batched_bf16(const void* descs, const int* chunk_map, int nchunks, void* stream);
/* fp32 image -> bf16 im2col matrix of the stem, Kpad a multiple of 64 */
int sacv_im2col_stem_bf16(const float* x, void* col, int Nb, int Cin, int H, int W, long long sN, long long sC,
                          long long sH, long long sW, int Ho, int Wo, int KH, int KW, int stride, int pad, int Kpad,
                          void* stream);
int sacv_bn_apply_bf16(const void* y, const float* mean, const float* invstd_or_var, const float* gamma,
                       const float* beta, const void* residual, void* z, long long numel, int C, int relu,
                       int stat_is_var, float eps, void* stream);
int sacv_bn_train_bwd_bf16(const void* dz, const void* zmask, const void* y, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, void* dy, void* g_out, float* dgamma, float* dbeta,
                           int R, int C, int accumulate, float* workspace, void* stream);
/* bf16 twins of sacv_conv2d_gemm_bnbwd / sacv_bn_train_bwd_partial (round 4): one data-gradient launch (mode 1, stride 1:
 * the data gradient of the NEXT layer in forward order) whose bf16 output dz [M][N] is the upstream gradient of the BatchNorm (+ ReLU) layer
 * with input bn_y also writes that layer's backward first-stage sums, stats[ceil(M / 128)][2][N] = (sum g, sum g * xhat)
 * over the rounded values it stores; flags SACV_CF_BIAS / SACV_CF_ACCUM.  Mask from bn_z, or recomputed from y (bn_gamma,
 * bn_beta) when bn_z is NULL, no mask when both are NULL. */
int sacv_conv2d_gemm_bf16_bnbwd(const void* src, void* dst, int Nb, int Hs, int Ws, int Hd, int Wd, const void* w,
                                const float* bias, int Cs, int N, int KH, int KW, int stride, int pad, int mode,
                                int flags, float* stats, const void* bn_y, const void* bn_z, const float* bn_mean,
                                const float* bn_invstd, const float* bn_gamma, const float* bn_beta, void* stream);
int sacv_bn_train_bwd_partial_bf16(const void* dz, const void* zmask, const void* y, const float* mean,
                                   const float* invstd, const float* gamma, const float* beta, void* dy, void* g_out,
                                   float* dgamma, float* dbeta, int R, int C, int accumulate, const float* partial,
                                   int nparts, float* workspace, void* stream);
int sacv_maxpool_fwd_bf16(const void* x, void* y, unsigned char* idx, int Nb, int H, int W, int C, int Ho, int Wo,
                          int k, int s, int p, void* stream);
int sacv_maxpool_bwd_bf16(const void* dy, const unsigned char* idx, void* dx, int Nb, int H, int W, int C, int Ho,
                          int Wo, int k, int s, int p, void* stream);
int sacv_avgpool_fwd_bf16(const void* x, void* out, int N, int HW, int C, void* stream);
int sacv_avgpool_bwd_bf16(const void* dy, void* dx, int N, int HW, int C, void* stream);

/* ---- ViT-B/16 MAE pre-training step (BASELINE config #5).  simpleAICV/masked_image_modeling/models/
 *      vit_mae.py:25-461, simpleAICV/classification/backbones/vit.py:20-160, masked_image_modeling/losses.py:11-23.
 *      The Linear layers (patch embedding, qkv, proj, fc1, fc2, encoder_to_decoder, pred) run as 1x1 problems of
 *      sacv_conv2d_gemm_bf16 / sacv_conv2d_wgrad_bf16; these are the kernels around them.  Index tensors are int64
 *      (torch.argsort / torch.gather), the residual stream is fp32, GEMM operands bf16. --------------------------- */
/* random_masking (vit_mae.py:187-215): shuffle = argsort(noise[b]), restore = argsort(shuffle), keep = shuffle[:keep],
 * mask = restore >= keep.  noise [B][n] fp32 (torch.rand in the reference), n <= 1024. */
int sacv_mae_random_masking(const float* noise, int B, int n, int keep_len, long long* shuffle_ids,
                            long long* restore_ids, long long* keep_ids, float* mask, void* stream);
/* rows[b*keep + j][(c*P + py)*P + px] bf16 = patch ids[b][j] of image b (ids NULL: every patch in order); the
 * P x P / stride-P PatchEmbeddingBlock convolution is a GEMM over these rows, only the KEPT patches are embedded */
int sacv_mae_patchify(const float* img, const long long* ids, void* rows, int B, int C, int H, int W, int P, int keep,
                      long long sN, long long sC, long long sH, long long sW, void* stream);
/* images_to_patch (vit_mae.py:421-433) [+ the collater's per-patch normalisation when norm != 0] -> fp32 [B][L][P*P*C] */
int sacv_mae_patch_labels(const float* img, float* out, int B, int C, int H, int W, int P, int norm, long long sN,
                          long long sC, long long sH, long long sW, void* stream);
/* encoder input: x[b][0] = cls + pos[0]; x[b][1+j] = emb[b*keep+j] + pos[1 + keep_ids[b][j]]  (fp32 out) */
int sacv_mae_encoder_tokens(const void* emb, const float* cls, const float* pos, const long long* keep_ids, float* x,
                            int B, int keep, int E, void* stream);
int sacv_mae_encoder_tokens_bwd(const float* dx, void* demb, int B, int keep, int E, void* stream);
/* decoder input (vit_mae.py:343-360): mask tokens appended, un-shuffled by restore_ids, + pos */
int sacv_mae_decoder_tokens(const void* y, const float* mask_token, const float* pos, const long long* restore_ids,
                            float* x, int B, int keep, int n, int D, void* stream);
/* dy bf16 [B][keep+1][D]; gmask fp32 [B*(n-keep)][D]: the rows whose column sum is d mask_token */
int sacv_mae_decoder_tokens_bwd(const float* dx, const long long* shuffle_ids, void* dy, float* gmask, int B, int keep,
                                int n, int D, void* stream);
/* nn.LayerNorm over the last dim: x fp32 [R][C] -> y bf16; mean / rstd fp32 [R] saved for the backward */
int sacv_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int R,
                       int C, float eps, void* stream);
long long sacv_layernorm_bwd_workspace_floats(int R, int C);
/* rows actually written of the partial buffer [rows][2C] (dbeta | dgamma) when the second stage is deferred
 * (dgamma = dbeta = NULL) to sacv_colsum_finalize_multi */
int sacv_layernorm_bwd_partial_rows(int R);
int sacv_layernorm_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                       float* dx, float* dgamma, float* dbeta, int R, int C, int accumulate_dx, int accumulate_params,
                       float* workspace, void* stream);
/* nn.GELU() (exact erf form) on bf16 tensors */
int sacv_gelu_fwd(const void* x, void* y, long long n, void* stream);
int sacv_gelu_bwd(const void* dy, const void* x, void* dx, long long n, void* stream);
long long sacv_colsum_bf16_workspace_floats(int R, int C);
/* rows actually written of the partial buffer [rows][C] when out = NULL (second stage deferred) */
int sacv_colsum_bf16_partial_rows(int R);
int sacv_colsum_bf16(const void* x, float* out, int R, int C, int accumulate, float* workspace, void* stream);
/* MultiHeadAttention core (vit.py:73-86): qkv bf16 [B][n][3][H][d] -> out bf16 [B][n][H*d];
 * softmax(bf16(QK^T) * scale) in fp32, probabilities cast to bf16 for the product with V (autocast's dtypes) */
int sacv_attention_fwd(const void* qkv, void* out, int B, int n, int H, int d, float scale, void* stream);
int sacv_attention_bwd(const void* qkv, const void* dout, void* dqkv, int B, int n, int H, int d, float scale,
                       void* stream);
/* the same block on the bf16 matrix cores (csrc/attention.hip: S^T = K Q^T, the probability tile in the accumulator
 * registers is the next MFMA's operand; two-phase deterministic backward from the saved log-sum-exp lse [B][H][n]);
 * covers head dims 32 / 64 and up to 224 tokens -- the VALU kernels above are the general form */
int sacv_attention_mfma_supported(int n, int d);
int sacv_attention_fwd_mfma(const void* qkv, void* out, float* lse, int B, int n, int H, int d, float scale,
                            void* stream);
int sacv_attention_bwd_mfma(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B,
                            int n, int H, int d, float scale, void* stream);
long long sacv_mae_mse_workspace_floats(int B, int n);
/* MSELoss(pred[:, 1:], label, mask) (losses.py:11-23); pred bf16 [B][n+1][K] (row 0 = cls token, dropped),
 * label fp32 [B][n][K], mask fp32 [B][n]; dpred (nullable) bf16 = d loss / d pred * grad_scale */
int sacv_mae_mse_loss(const void* pred, const float* label, const float* mask, void* dpred, float* loss_out, int B,
                      int n, int K, float grad_scale, float* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SACV_HIP_H */
