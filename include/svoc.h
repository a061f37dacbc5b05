/*
 * svoc.h — C ABI of libsvoc_hip.so: the MI355X (gfx950) implementation of the
 * SMART-Vocoder inference path.
 *
 * The reference (SMART-TTS/SMART-Vocoder) is pure Python/PyTorch and has no
 * FFI of its own; the drop-in boundary is its Python surface (models.py /
 * modules.py / transforms.py).  Each entry point below is what a ctypes
 * binding of one reference class or function would call; the reference
 * interface it replaces is cited as file:line (relative to the reference repo).
 * `smart-vocoder_amd/_native.py` is that binding; INTEGRATION.md shows the stub
 * a maintainer of the reference would add.
 *
 * Conventions
 *   - every tensor is fp32, device memory, contiguous NCW unless a stride is
 *     given; lengths are int64 device memory;
 *   - every function returns 0 on success or a negative svoc_status and never
 *     throws or exits;
 *   - blocking behaviour: `*_forward` / `svoc_synth_infer` and the single ops
 *     only enqueue work, with ONE exception: the first call at a shape larger
 *     than any seen before grows the handle's workspace, which waits for the
 *     device (hipDeviceSynchronize) before hipFree/hipMalloc; call
 *     svoc_synth_reserve(h, B, T) once with the largest shape to make every later
 *     infer allocation-free.  `*_create` blocks while weights are folded and
 *     repacked; `*_destroy` waits for the device before freeing; svoc_conv1d /
 *     svoc_conv_transpose1d (self-contained test entry points that pack their
 *     weights on the fly) block until their result is ready;
 *   - work is enqueued on the caller's hipStream_t (passed as void*); handles
 *     are per-device and not thread-safe; the caller selects the device
 *     (hipSetDevice) before create;
 *   - the library owns packed weights and its activation workspace (grow-only
 *     hipMalloc), the caller owns all inputs and outputs;
 *   - state-dict tables: weights are handed over as an array of svoc_tensor
 *     named exactly like the reference's state_dict keys below a prefix
 *     (e.g. "in_layers.0.weight_v"); weight-norm pairs (weight_g, weight_v) are
 *     folded on the device at create time (reference: torch weight_norm at
 *     modules.py:128,135,145,191-206, models.py:125; removal models.py:162-167).
 */
#ifndef SVOC_H
#define SVOC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum svoc_status {
  SVOC_OK = 0,
  SVOC_ERR_INVALID_ARG = -1,
  SVOC_ERR_MISSING_TENSOR = -2,
  SVOC_ERR_SHAPE = -3,
  SVOC_ERR_HIP = -4,
  SVOC_ERR_UNSUPPORTED = -5,
  SVOC_ERR_NOMEM = -6
} svoc_status;

typedef struct svoc_tensor {
  const char* name;    /* state_dict key below the module prefix */
  const float* data;   /* device pointer, fp32, contiguous */
  int32_t ndim;
  int64_t shape[4];
} svoc_tensor;

/* Library / error reporting ------------------------------------------------ */
int svoc_abi_version(void);
const char* svoc_last_error(void);       /* thread-local message of the last failure */
const char* svoc_build_arch(void);       /* "gfx950" */
/* Launch statistics since the last reset: GEMM-family kernel launches (conv_mfma / conv_group / resblock_fused /
 * wn_layer_fused; a grouped or fused launch counts once), their algorithmic FLOPs (2*MAC), launches of the small
 * memory-bound kernels, and the number of CONVOLUTIONS those GEMM launches computed (a group launch = 3, a fused
 * ResBlock iteration or WN layer = 2) — used by bench.py's roofline block. */
int svoc_stats_reset(void);
int svoc_stats_get(int64_t* conv_launches, double* conv_flops, int64_t* other_launches);
int64_t svoc_stats_convolutions(void);
/* 2 x the multiply-adds the matrix pipe ISSUED for those launches: equal to conv_flops for direct-form kernels; per Winograd
 * form, as a share of the direct form for k = 3 / 7 / 11: F(4,3) (k = 3) 1/2;
 * F(4,4) (default for k = 7 / 11 of every F(4,3)-eligible shape; four-tap groups in seven products) 3.5/7, 5.25/11;
 * F(2,3) (fall-back) 2/3, 5/7, 8/11; F(4,2) (upsamplers, k = 2 stride) 5/8; F(2,5) (WN in_layers, k = 5) 3/5.
 * bench.py's roofline.achieved / frac is this / time (/ peak). */
double svoc_stats_executed_flops(void);

/* Kernel variants (tile shape, K split, Winograd or direct form, fused or unfused WN layer, MRF launch plan) are chosen
 * from the launch size, and variants differ in summation order: by default one utterance alone and the same utterance
 * inside a batch agree to fp32 rounding (<= 2e-6), not bit for bit.  svoc_set_variant_batch(n) makes every such choice
 * as if the batch held n utterances (n <= 0: the real batch, the default; env SVOC_VARIANT_BATCH presets it).  A rank
 * that runs a B/G-utterance shard with n = B produces exactly the bits of a single process running all B (SURVEY.md
 * 8e "8-GPU output == 1-GPU output bitwise"); used by parallel.infer_sharded(bitwise=True).  Process-global; returns the
 * previous value.  Replaces nothing in the reference (it has one code path per op: torch's). */
int svoc_set_variant_batch(int n);
/* Two kernels of the path are PERSISTENT launches whose workgroups wait for each other (a whole WN stack per launch: csrc/wn_stack.hip,
 * csrc/wn_mesh.hip).  They are taken only while the runtime's occupancy calculator says the whole grid can be resident on the device
 * (hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs, asked once per kernel and device).  Alone on a GPU - one process per GPU is the deployment
 *  - nothing waits for long; when other work shares the GPU two such launches can hold each other's CUs, so every wait is
 * bounded by wall time: SVOC_PERSIST_TIMEOUT_MS, default 2000.  A workgroup that gives up
 *   (1) turns its part of THAT call's result into NaN (the stack's output rows of its tile, and through them the call's waveform: the reference's
 *       WN.forward, modules.py:148-176, cannot return a wrong finite tensor, and neither can this one),
 *   (2) raises a host-visible word: the NEXT svoc_wn_forward / svoc_synth_infer - or svoc_check_async_error(), which needs no synchronisation and
 *       can be called right behind the caller's own stream synchronisation of the affected call - fails with SVOC_ERR_HIP and clears the word,
 *   (3) and from that report on the process runs one launch per WN layer (same function, bit-identical for the stack launch); captured plans are
 *       captured again.
 * Nothing in the reference corresponds (torch launches one kernel per op). */
int svoc_check_async_error(void);             /* ABI 5 */
/* *disabled = 1 once (3) has happened; *timeout_ms = the bound in force.  Either pointer may be NULL.  ABI 6 */
int svoc_persist_state(int* disabled, int* timeout_ms);
/* diagnostics: raises that word from the host, exactly as a workgroup that gave up would (tests of the reporting path) */
int svoc_debug_raise_async_error(void);
/* diagnostics (ABI 6): fault_tile >= 0 makes the workgroup of that 32-column tile withhold the flags its neighbours wait for in every persistent
 * launch from now on (-1: off); timeout_ms > 0 overrides SVOC_PERSIST_TIMEOUT_MS (<= 0: the configured bound); reenable != 0 undoes (3). */
int svoc_debug_persist_control(int fault_tile, int timeout_ms, int reenable);

/* Diagnostics: bracket every convolution launch with HIP events and aggregate by layer shape. */
int svoc_profile_enable(int on);
int svoc_profile_report(char* buf, int buflen);
/* Diagnostics: run lrelu->Conv1d(C->C,k,d)[+residual] twice with per-workgroup cycle stamps; out4 = mean cycles of
 * {staging of the first stage, MFMA (+later stages), epilogue} and the span of the last launch. */
/* diagnostics: [workgroup][8] int64 device buffer for resblock_fused_kernel's phase cycle stamps (NULL = off) */
int svoc_debug_set_stamp_buffer(void* buf);
int svoc_debug_conv_timing(void* stream, const float* x, const float* weight, const float* bias, const float* residual, float* y,
                           int B, int C, int L, int kernel_size, int dilation, double* out4);

/* ---- modules.WN (modules.py:111-185) ------------------------------------- */
typedef struct svoc_wn svoc_wn;
/* tensors: in_layers.{i}.{bias,weight_g,weight_v}, res_skip_layers.{i}.*, cond_layer.* if gin_channels>0 */
int svoc_wn_create(svoc_wn** out, int hidden_channels, int kernel_size, int dilation_rate, int n_layers,
                   int gin_channels, const svoc_tensor* tensors, int n_tensors, const char* prefix);
/* WN.forward(x, x_mask, g) (modules.py:148-176): x [B,H,T], x_mask [B,1,T], g NULL or [B,gin,Tg] (Tg==1 or T),
 * out [B,H,T]. */
int svoc_wn_forward(svoc_wn* h, void* stream, const float* x, const float* x_mask, const float* g, int g_T,
                    float* out, int B, int T);
void svoc_wn_destroy(svoc_wn* h);

/* ---- modules.ResBlock1 / ResBlock2 (modules.py:187-256) ------------------- */
typedef struct svoc_resblock svoc_resblock;
/* kind 1: convs1.{0..n-1}.*, convs2.{0..n-1}.* ; kind 2: convs.{0..n-1}.* ; n = n_dilations */
int svoc_resblock_create(svoc_resblock** out, int kind, int channels, int kernel_size, const int* dilations,
                         int n_dilations, const svoc_tensor* tensors, int n_tensors, const char* prefix);
/* forward(x, x_mask=None): x,y [B,C,L]; x_mask NULL or [B,1,L] */
int svoc_resblock_forward(svoc_resblock* h, void* stream, const float* x, const float* x_mask, float* y, int B, int L);
void svoc_resblock_destroy(svoc_resblock* h);

/* ---- modules.ResidualCouplingLayer (modules.py:298-343) ------------------- */
typedef struct svoc_coupling svoc_coupling;
/* tensors: pre.{weight,bias}, enc.<WN tensors>, post.{weight,bias} */
int svoc_coupling_create(svoc_coupling** out, int channels, int hidden_channels, int kernel_size, int dilation_rate,
                         int n_layers, int gin_channels, int mean_only, const svoc_tensor* tensors, int n_tensors,
                         const char* prefix);
/* forward(x, x_mask, g, reverse): x,y [B,C,T]; logdet [B] written when reverse==0 (may be NULL) */
int svoc_coupling_forward(svoc_coupling* h, void* stream, const float* x, const float* x_mask, const float* g, int g_T,
                          int reverse, float* y, float* logdet, int B, int T);
void svoc_coupling_destroy(svoc_coupling* h);

/* ---- models.ResidualCouplingBlock (models.py:50-80) incl. modules.Flip (modules.py:270-277) */
typedef struct svoc_flow svoc_flow;
/* tensors: flows.{0,2,4,..}.<coupling tensors>; Flips are folded into channel-permuted weights */
int svoc_flow_create(svoc_flow** out, int channels, int hidden_channels, int kernel_size, int dilation_rate,
                     int n_layers, int n_flows, int gin_channels, const svoc_tensor* tensors, int n_tensors,
                     const char* prefix);
int svoc_flow_forward(svoc_flow* h, void* stream, const float* x, const float* x_mask, const float* g, int g_T,
                      int reverse, float* y, int B, int T);
void svoc_flow_destroy(svoc_flow* h);

/* ---- models.Generator (models.py:115-167) --------------------------------- */
typedef struct svoc_generator svoc_generator;
typedef struct svoc_generator_config {
  int32_t initial_channel;
  int32_t resblock_kind;                 /* 1 or 2 ("resblock": "1"|"2") */
  int32_t n_kernels;                     /* len(resblock_kernel_sizes) <= 8 */
  int32_t resblock_kernel_sizes[8];
  int32_t n_dilations[8];
  int32_t resblock_dilation_sizes[8][8];
  int32_t n_upsamples;                   /* <= 8 */
  int32_t upsample_rates[8];
  int32_t upsample_kernel_sizes[8];
  int32_t upsample_initial_channel;
  int32_t gin_channels;
} svoc_generator_config;
/* tensors: conv_pre.*, ups.{i}.*, resblocks.{j}.*, conv_post.weight, cond.* if gin_channels>0 */
int svoc_generator_create(svoc_generator** out, const svoc_generator_config* cfg, const svoc_tensor* tensors,
                          int n_tensors, const char* prefix);
/* Generator.forward(x, g): x [B,initial_channel,T] with row stride x_ld (>=T), optional input mask [B,1,*]
 * (infer passes z*x_mask, models.py:338), g NULL or [B,gin,1]; out [B,1,T*prod(upsample_rates)]. */
int svoc_generator_forward(svoc_generator* h, void* stream, const float* x, int x_ld, int64_t x_bs,
                           const float* in_mask, int64_t in_mask_bs, const float* g, float* out, int B, int T);
void svoc_generator_destroy(svoc_generator* h);

/* ---- models.SynthesizerTrn.infer (models.py:266-339) ---------------------- */
typedef struct svoc_synth svoc_synth;
typedef struct svoc_synth_config {
  int32_t n_mel;                 /* 80: MelEncoder.pre_enc in-channels (models.py:32) */
  int32_t inter_channels;        /* 192 */
  int32_t hidden_channels;       /* 192 */
  int32_t enc_n_layers;          /* 16  (models.py:308) */
  int32_t enc_kernel_size;       /* 5 */
  int32_t enc_dilation_rate;     /* 1 */
  int32_t flow_n_layers;         /* 8   (models.py:314) */
  int32_t flow_kernel_size;      /* 5 */
  int32_t flow_dilation_rate;    /* 1 */
  int32_t flow_n_flows;          /* 4 */
  int32_t gin_channels;          /* 256 (conditioning weights exist but infer passes g=None, models.py:332) */
  svoc_generator_config dec;
} svoc_synth_config;
/* tensors: the reference state_dict (enc_p.*, flow.*, dec.*; enc_q.* ignored) */
int svoc_synth_create(svoc_synth** out, const svoc_synth_config* cfg, const svoc_tensor* tensors, int n_tensors);
/* infer(x, x_lengths, noise_scale, max_len) with the randn_like draw (models.py:336) supplied by the caller:
 *   mel [B,n_mel,T]; lengths int64 [B]; eps [B,inter,T] or NULL (then noise_scale must be 0);
 *   max_len <= 0 means None; outputs (any may be NULL except o):
 *   o [B,1,min(T,max_len)*hop], x_mask [B,1,T], z,z_p,m_p,logs_p [B,inter,T]. */
int svoc_synth_infer(svoc_synth* h, void* stream, const float* mel, const int64_t* lengths, const float* eps,
                     float noise_scale, int max_len, float* o, float* x_mask, float* z, float* z_p, float* m_p,
                     float* logs_p, int B, int T);
int64_t svoc_synth_workspace_bytes(svoc_synth* h, int B, int T);
/* Pre-sizes every workspace of the path for batches up to [B, T] (may block, see "blocking behaviour"); afterwards
 * svoc_synth_infer at that or any smaller shape neither allocates nor synchronises.  Batches of up to 32768 frames (B*T,
 * SVOC_GRAPH_MAX_FRAMES) are replayed from a hipGraph captured on their second call (SVOC_GRAPH=0 disables). */
int svoc_synth_reserve(svoc_synth* h, int B, int T);
int svoc_synth_hop(svoc_synth* h);         /* prod(upsample_rates) */
/* Captured-plan bookkeeping of a handle: out5 = {live plans, captures so far, plans evicted, evictions that had to wait
 * (hipEventSynchronize on the evicted plan's own completion event), plans retired but still executing}.  A shape earns
 * a plan on its SVOC_GRAPH_MIN_SEEN-th call (default 2; first sights only bump a counter); at most 32 plans live, the
 * least recently used makes room, and eviction never synchronises the device nor frees device memory. */
int svoc_synth_plan_stats(svoc_synth* h, int64_t* out5);
void svoc_synth_destroy(svoc_synth* h);

/* ---- models.PosteriorEncoder (models.py:83-112): not on the infer path; enc_q of training / voice conversion ---- */
typedef struct svoc_posterior svoc_posterior;
/* tensors: pre.{weight,bias}, enc.<WN tensors>, proj.{weight,bias} */
int svoc_posterior_create(svoc_posterior** out, int in_channels, int out_channels, int hidden_channels, int kernel_size,
                          int dilation_rate, int n_layers, int gin_channels, const svoc_tensor* tensors, int n_tensors,
                          const char* prefix);
/* forward(x, x_lengths, g) (models.py:105-112): x [B,in,T], lengths [B] int64, g NULL or [B,gin,Tg], eps [B,out,T] = the
 * reference's torch.randn_like(m) draw; writes z = (m + eps*exp(logs))*mask, m, logs [B,out,T] and x_mask [B,1,T] (may be NULL) */
int svoc_posterior_forward(svoc_posterior* h, void* stream, const float* x, const int64_t* lengths, const float* g, int g_T,
                           const float* eps, float* z, float* m, float* logs, float* x_mask, int B, int T);
void svoc_posterior_destroy(svoc_posterior* h);


/* ---- models.MelEncoder (models.py:15-47) as a standalone module (inside svoc_synth_infer it is part of the one call) */
typedef struct svoc_mel_encoder svoc_mel_encoder;
/* tensors: pre_enc.{weight,bias}, encoder.<WN tensors>, proj.{weight,bias} */
int svoc_mel_encoder_create(svoc_mel_encoder** out, int n_mel, int out_channels, int hidden_channels, int kernel_size,
                            int dilation_rate, int n_layers, int gin_channels, const svoc_tensor* tensors, int n_tensors,
                            const char* prefix);
/* forward(x, x_lengths) (models.py:35-47; g is discarded there): x [B,n_mel,T], lengths int64 [B] ->
 * x_out [B,hidden,T] (WN output, may be NULL), m, logs [B,out,T], x_mask [B,1,T] (may be NULL) */
int svoc_mel_encoder_forward(svoc_mel_encoder* h, void* stream, const float* x, const int64_t* lengths, float* x_out,
                             float* m, float* logs, float* x_mask, int B, int T);
void svoc_mel_encoder_destroy(svoc_mel_encoder* h);

/* ---- modules.LayerNorm.forward (modules.py:20-32): F.layer_norm over the channel dim of x [B,C,T]; y [B,C,T] */
int svoc_layer_norm(void* stream, const float* x, const float* gamma, const float* beta, float eps, float* y, int B,
                    int C, int T);

/* ---- modules.DDSConv (modules.py:70-108) ---------------------------------- */
typedef struct svoc_dds svoc_dds;
/* tensors: convs_sep.{i}.{weight,bias}, convs_1x1.{i}.{weight,bias}, norms_1.{i}.{gamma,beta}, norms_2.{i}.* */
int svoc_dds_create(svoc_dds** out, int channels, int kernel_size, int n_layers, const svoc_tensor* tensors,
                    int n_tensors, const char* prefix);
/* forward(x, x_mask, g): x,y [B,C,T]; g NULL or [B,C,g_T] with g_T == T or 1 (broadcast over time) */
int svoc_dds_forward(svoc_dds* h, void* stream, const float* x, const float* x_mask, const float* g, int g_T, float* y,
                     int B, int T);
void svoc_dds_destroy(svoc_dds* h);

/* ---- modules.ConvFlow (modules.py:346-390) -------------------------------- */
typedef struct svoc_convflow svoc_convflow;
/* tensors: pre.*, convs.<DDSConv tensors>, proj.* */
int svoc_convflow_create(svoc_convflow** out, int in_channels, int filter_channels, int kernel_size, int n_layers,
                         int num_bins, float tail_bound, const svoc_tensor* tensors, int n_tensors, const char* prefix);
/* forward(x, x_mask, g, reverse) (modules.py:363-390): g NULL or [B,filter_channels,g_T] (g_T == T or 1), handed to
 * DDSConv (modules.py:366) */
int svoc_convflow_forward(svoc_convflow* h, void* stream, const float* x, const float* x_mask, const float* g, int g_T,
                          int reverse, float* y, float* logdet, int B, int T);
void svoc_convflow_destroy(svoc_convflow* h);

/* ---- mel_processing.spectrogram_torch / spec_to_mel_torch / mel_spectrogram_torch (mel_processing.py:51-112) ------ */
typedef struct svoc_melspec svoc_melspec;
/* n_fft must be a multiple of hop_length and win_length == n_fft (the reference uses 1024/256/1024); fmax <= 0 means
 * sampling_rate/2 (the config's mel_fmax: null).  The mel basis restates librosa 0.8.0 filters.mel (Slaney). */
int svoc_melspec_create(svoc_melspec** out, int n_fft, int hop_length, int win_length, int n_mels, int sampling_rate,
                        double fmin, double fmax);
int svoc_melspec_frames(svoc_melspec* h, int64_t n_samples);      /* frames produced for a waveform of n_samples */
/* y [B, n_samples] -> spec [B, n_fft/2+1, frames] = sqrt(re^2 + im^2 + 1e-6) (reflect pad (n_fft-hop)/2, center=False) */
int svoc_melspec_spectrogram(svoc_melspec* h, void* stream, const float* y, int B, int n_samples, float* spec);
/* ABI 4: the same with torch.stft's center flag (mel_processing.py:66-67 forwards it): center != 0 frames the signal after a
 * second reflect padding of n_fft/2 on both sides, as torch.stft(center=True, pad_mode='reflect') does; center == 0 is the call above */
int svoc_melspec_frames_center(svoc_melspec* h, int64_t n_samples, int center);
int svoc_melspec_spectrogram_center(svoc_melspec* h, void* stream, const float* y, int B, int n_samples, int center, float* spec);
/* spec [B, n_fft/2+1, F] -> mel [B, n_mels, F] = log(clamp(mel_basis @ spec, 1e-5)) */
int svoc_melspec_mel(svoc_melspec* h, void* stream, const float* spec, int B, int n_frames, float* mel);
int svoc_mel_filterbank(int sampling_rate, int n_fft, int n_mels, double fmin, double fmax, float* out_host);
void svoc_melspec_destroy(svoc_melspec* h);

/* ---- transforms.piecewise_rational_quadratic_transform (transforms.py:12-193), tails='linear' or none */
/* inputs [n]; unnormalized widths/heights [n,num_bins]; derivatives [n,num_bins-1] (linear tails) or
 * [n,num_bins+1] (tails==0, domain [0,1]); min_bin_width/min_bin_height/min_derivative as transforms.py:20-22
 * (defaults 1e-3); outputs, logabsdet [n]. */
int svoc_rq_spline(void* stream, const float* inputs, const float* unnorm_widths, const float* unnorm_heights,
                   const float* unnorm_derivs, int64_t n, int num_bins, int inverse, int linear_tails,
                   float tail_bound, float min_bin_width, float min_bin_height, float min_derivative, float* outputs,
                   float* logabsdet);

/* ---- single ops (unit-testable pieces of the path) ------------------------- */
/* commons.sequence_mask (commons.py:121-125) as float [B,1,T] */
int svoc_sequence_mask(void* stream, const int64_t* lengths, float* mask, int B, int T);
/* commons.fused_add_tanh_sigmoid_multiply (commons.py:100-107): a,b [B,2H,T] -> acts [B,H,T] */
int svoc_fused_add_tanh_sigmoid_multiply(void* stream, const float* a, const float* b, float* acts, int B, int H, int T);
/* modules.Flip (modules.py:270-277): y[b][c] = x[b][C-1-c]; x,y [B,C,T] */
int svoc_flip_channels(void* stream, const float* x, float* y, int B, int C, int T);
/* torch.nn.utils.remove_weight_norm (models.py:162-167, modules.py:178-184): w = g * v / ||v||, norm over all
 * dims but 0; v,w [d0, inner], g [d0] */
int svoc_fold_weight_norm(void* stream, const float* weight_v, const float* weight_g, float* weight, int64_t d0,
                          int64_t inner);
/* F.leaky_relu -> (weight-normed) Conv1d with "same" zero padding (commons.get_padding, commons.py:14-15)
 * [-> + residual]: the decoder's unit of work (modules.py:212-218).  weight_g may be NULL (plain conv).
 * x [B,Cin,L], y [B,Cout,L], residual NULL or [B,Cout,L]; pre_slope 1.0 = no activation. */
int svoc_conv1d(void* stream, const float* x, const float* weight_v, const float* weight_g, const float* bias,
                const float* residual, float* y, int B, int Cin, int Cout, int L, int kernel_size, int dilation,
                float pre_slope);
/* The same operation for kernel_size 3 / 7 / 11, dilation 1 / 3 / 5, channel counts multiples of 32 (Cin >= 64, or Cin = Cout = 32)
 * and L % 4 == 0, computed in Winograd form: F(4,3) (csrc/conv_wino4.hip: Cout in 128- / 64-row blocks with an even number of
 * 32-channel chunks, or the single 32 x 32 block of the last MRF stage) - what every ResBlock convolution of the decoder runs -
 * with k = 7 / 11 of EVERY F(4,3)-eligible shape (128-, 64- and 32-row blocks alike) in F(4,4) form (csrc/conv_wino4.h,
 * instantiated in csrc/conv_wino44_r{4,2,1}.hip);
 * F(2,3) (csrc/conv_wino.hip) for the other shapes and with SVOC_WINO_F4=0.  svoc_stats_executed_flops tells which form ran.
 * (Inside the decoder the grouped launches additionally hand tensors over window-major between a dilated convolution and the one
 * behind it, and merge the three chains' last convolutions: csrc/conv_wino4.hip, csrc/conv_wino4_acc.hip - not reachable from here.)
 * Unit-test entry; returns SVOC_ERR_UNSUPPORTED for other shapes. */
int svoc_conv1d_winograd(void* stream, const float* x, const float* weight_v, const float* weight_g, const float* bias,
                         const float* residual, float* y, int B, int Cin, int Cout, int L, int kernel_size, int dilation,
                         float pre_slope);
/* F.leaky_relu -> weight-normed ConvTranspose1d(k, stride, padding=(k-stride)//2) (models.py:125-127, 147-148):
 * x [B,Cin,L], weight_v [Cin,Cout,k], weight_g [Cin,1,1] or NULL, y [B,Cout,L*stride].  k = 2 * stride with stride 8 or 2,
 * Cin % 64 == 0, Cout * stride % 64 == 0, L % 4 == 0 and at least half as many tiles as CUs: Winograd F(4,2) over the polyphase
 * filters (csrc/convt_wino.hip, what the decoder's upsamplers run); otherwise the direct polyphase GEMM (csrc/conv_mfma.hip). */
int svoc_conv_transpose1d(void* stream, const float* x, const float* weight_v, const float* weight_g,
                          const float* bias, float* y, int B, int Cin, int Cout, int L, int kernel_size, int stride,
                          float pre_slope);

#ifdef __cplusplus
}
#endif
#endif /* SVOC_H */
