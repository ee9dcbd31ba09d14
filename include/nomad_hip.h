/*
 * nomad_hip.h - C ABI of libnomad_hip.so, the MI355X (gfx950) NOMAD scoring engine.
 *
 * The reference (alessandroragano/nomad) has no FFI boundary of its own: its hot path is Python
 * calling into fairseq/PyTorch/SciPy.  Each entry point below replaces one such call site
 * (paths relative to /root/reference):
 *
 *   nomad_create / nomad_destroy   Nomad.__init__ model construction + load_state_dict
 *                                  (src/nomad_audio/nomad.py:57-71)
 *   nomad_embed                    TripletModel.forward (nomad.py:224-231) and
 *                                  LossNetLayers.forward (nomad.py:243-258): wav2vec 2.0 BASE
 *                                  backbone (fairseq Wav2Vec2Model, call sites nomad.py:226,245)
 *                                  + mean/ReLU/Linear/L2-normalise head
 *   nomad_pairwise                 scipy cdist + np.mean(axis=1) (nomad.py:108-111)
 *   nomad_wav_probe / _read_rows   torchaudio.load + channel mean + Resample of load_processing (nomad.py:196-205)
 *   nomad_l1_loss                  NomadLoss.forward (nomad.py:267-282)
 *
 * Conventions: every function returns 0 on success or a negative nomad_status; nothing throws.
 * All `dev` pointers are device (HBM) pointers owned by the caller; `host` pointers are host
 * memory.  Compute entry points are asynchronous on the caller's hipStream_t and never allocate:
 * scratch comes from the caller-provided workspace (size from nomad_workspace_bytes).  A context
 * is bound to one device and is not re-entrant.  The library has no CPU fallback: without a
 * gfx950 device nomad_create fails with NOMAD_ERR_NO_DEVICE.
 */
#ifndef NOMAD_HIP_H
#define NOMAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NOMAD_NUM_LAYERS 12
#define NOMAD_EMBED_DIM 768
#define NOMAD_EMB_DIM 256

typedef enum nomad_status {
    NOMAD_OK = 0,
    NOMAD_ERR_INVALID = -1,   /* bad argument */
    NOMAD_ERR_NO_DEVICE = -2, /* no usable gfx950 device */
    NOMAD_ERR_HIP = -3,       /* a HIP runtime call failed; see nomad_last_error() */
    NOMAD_ERR_WORKSPACE = -4, /* workspace too small */
    NOMAD_ERR_IO = -5,        /* a file could not be opened / read (nomad_wav_*) */
    NOMAD_ERR_FORMAT = -6     /* not a RIFF/WAVE file, or an encoding nomad_wav_read_rows does not decode */
} nomad_status;

typedef struct nomad_ctx nomad_ctx;
typedef void* nomad_stream_t; /* hipStream_t */

/* One transformer encoder layer, fairseq TransformerSentenceEncoderLayer parameter names.
 * All matrices are row-major [out][in] exactly as torch.nn.Linear stores them. */
typedef struct nomad_layer_weights {
    const float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *o_w, *o_b; /* self_attn.{q,k,v,out}_proj */
    const float *ln1_w, *ln1_b;                                 /* self_attn_layer_norm */
    const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;                 /* fc1 [3072][768], fc2 [768][3072] */
    const float *ln2_w, *ln2_b;                                 /* final_layer_norm */
} nomad_layer_weights;

/* HOST pointers, fp32, in the layout of the reference checkpoint nomad_best_model.pt
 * (keys ssl_model.* / embedding_layer.1.*; nomad.py:63-65).  nomad_create copies and repacks
 * them; the caller may free them afterwards. */
typedef struct nomad_weights {
    const float* conv_w[7];          /* feature_extractor.conv_layers.i.0.weight: [512][1][10], 4x[512][512][3], 2x[512][512][2] */
    const float *gn_w, *gn_b;        /* feature_extractor.conv_layers.0.2 (GroupNorm 512) */
    const float *feat_ln_w, *feat_ln_b; /* layer_norm (512) */
    const float *proj_w, *proj_b;    /* post_extract_proj [768][512] */
    const float *pos_v, *pos_g, *pos_b; /* encoder.pos_conv.0 weight_v [768][48][128], weight_g [128], bias */
    const float *enc_ln_w, *enc_ln_b;   /* encoder.layer_norm */
    nomad_layer_weights layers[NOMAD_NUM_LAYERS];
    const float *emb_w, *emb_b;      /* embedding_layer.1 [256][768], [256] */
} nomad_weights;

/* ---- lifetime ---------------------------------------------------------------------------- */
int nomad_create(nomad_ctx** out, int device, const nomad_weights* host_weights);
void nomad_destroy(nomad_ctx* ctx);
const char* nomad_last_error(void);
const char* nomad_version(void);
/* Binary-interface number of this header.  It changes whenever an entry point's signature or a struct layout changes in a way an
 * already compiled caller would not survive; a caller compares nomad_abi_version() of the library it loaded with the
 * NOMAD_ABI_VERSION it was compiled against BEFORE any other call (nomad_amd/_lib.py does).  History: 2 -> 3 (round 5/6):
 * nomad_set_concurrent_parts(int) became nomad_set_concurrent_parts(nomad_ctx*, int) - a caller built against the old header
 * would pass its int where the context pointer goes. */
#define NOMAD_ABI_VERSION 3
int nomad_abi_version(void);
/* How this library was built, as a bit set.  NOMAD_BUILD_PACKED_FP32: the device code may contain packed-FP32 VALU
 * instructions (v_pk_fma_f32 ...), which on gfx950 can lose a product while a bf16 MFMA kernel of ANOTHER stream shares the
 * SIMD (DESIGN.md "The packed-FP32 hazard") - the shipped build has none, and a host layer must not co-schedule two forwards
 * of a context on two streams when this bit is set (nomad_amd.Engine switches its two-stream batch split off).  The library's
 * own 128 x 128 bf16 GEMM can still disturb third-party kernels that use packed FP32 on other streams.
 * NOMAD_BUILD_DIAG: libnomad_diag.so (every experimental instantiation and probe). */
#define NOMAD_BUILD_PACKED_FP32 1
#define NOMAD_BUILD_DIAG 2
int nomad_build_flags(void);
/* Scheduling hint: into how many parts, on as many streams, the host layer splits the batches it submits CONCURRENTLY (1 = one
 * forward at a time, the default).  Results never depend on it - every fp32 GEMM instantiation contracts k in the same order -
 * only the tile-shape choice does: next to another stream's kernels the 128 x 128 tiles' extra operand traffic costs more
 * (measured: the 256 x 128 / 128 x 128 price ratio that minimises the bench step is 1.08 with two concurrent halves, 1.03 with
 * one forward).  nomad_amd.Engine calls it with its split count.  Per context (round 5: it was process-wide state); like every
 * other call on a context it must not race with another host thread's call on the SAME context.  The library keeps no mutable
 * process-wide state that affects results (what is process-wide: the last-error text, the once-per-kernel LDS-attribute flags, the
 * diagnostic library's timeline buffer) and never reads the environment (tests/test_abi.py holds libnomad_hip.so to "imports no getenv").
 * Returns 0, or NOMAD_ERR_INVALID for a null context or parts < 1. */
int nomad_set_concurrent_parts(nomad_ctx* ctx, int parts);

/* ---- shapes ------------------------------------------------------------------------------ */
/* Encoder frames T for a clip of n_samples (conv stack (10,5),(3,2)x4,(2,2)x2); <=0 if too short. */
int nomad_num_frames(int n_samples);
/* Scratch bytes nomad_embed needs for a (B, n_samples) batch. */
int nomad_workspace_bytes(const nomad_ctx* ctx, int B, int n_samples, size_t* bytes);

/* ---- hot path ---------------------------------------------------------------------------- */
/*
 * Embed B clips of n_samples each.
 *   wav_dev     [B][n_samples] fp32 raw amplitude (nomad.py:225 squeezes (B,1,N) to this)
 *   head_w_dev/head_b_dev  optional override of the 768->256 head ([256][768],[256], device);
 *               NULL = the checkpoint's embedding_layer (TripletModel).  LossNetLayers has its own
 *               never-loaded embedding layer (nomad.py:238-241), which callers pass here.
 *   emb_dev     [B][256] fp32 unit-norm embeddings (out)
 *   layers_dev  optional [12][B][T][768] fp32 transformer layer outputs (out; nomad.py:248), or NULL
 */
int nomad_embed(nomad_ctx* ctx, const float* wav_dev, int B, int n_samples,
                const float* head_w_dev, const float* head_b_dev,
                float* emb_dev, float* layers_dev,
                void* workspace_dev, size_t workspace_bytes, nomad_stream_t stream);

/*
 * Embed B clips of DIFFERENT lengths in one launch sequence, with no padding in the arithmetic (the
 * reference's per-file loop, nomad.py:171-183, batched): results are bit-identical to per-clip nomad_embed
 * calls.  wav_dev is [B][stride] fp32 with clip b occupying the first lengths_host[b] samples of its row
 * (the rest of the row is never read); lengths_host is a HOST array.
 */
int nomad_workspace_bytes_ragged(const nomad_ctx* ctx, int B, const int* lengths_host, size_t* bytes);
int nomad_embed_ragged(nomad_ctx* ctx, const float* wav_dev, int B, int stride, const int* lengths_host,
                       const float* head_w_dev, const float* head_b_dev, float* emb_dev,
                       void* workspace_dev, size_t workspace_bytes, nomad_stream_t stream);

/*
 * Euclidean distance matrix + row means, computed in float64 in the difference form
 * sqrt(sum_k (a_k - b_k)^2) like scipy's cdist on float32 inputs promoted to double.
 *   deg_dev [Nd][256] fp32, ref_dev [Nr][256] fp32
 *   dist_dev  optional [Nd][Nr] float64 (out) or NULL;  mean_dev [Nd] float64 (out)
 * Scratch (row sums per 64-reference tile, 16 MB) is owned by the context, one block PER LAUNCH STREAM: calls on
 * different streams of one context may be in flight together; the first call on a stream the context has not seen
 * allocates that stream's block (the only entry point that may allocate after nomad_create).  That look-up is guarded by a
 * mutex in the context: host threads that each drive their own stream may call it concurrently.  (Every other entry point
 * that takes a context expects one driving host thread per context.)
 */
int nomad_pairwise(nomad_ctx* ctx, const float* deg_dev, int Nd, const float* ref_dev, int Nr,
                   double* dist_dev, double* mean_dev, nomad_stream_t stream);

/*
 * NomadLoss: sum_{i<12} mean|a_layers[i]-b_layers[i]| + mean|a_emb-b_emb|.
 *   a_layers_dev/b_layers_dev [12][B][T][768]; a_emb_dev/b_emb_dev [B][256]; loss_dev [1] fp32 (out)
 *   scratch_dev: >= nomad_l1_scratch_bytes() bytes of device scratch.
 */
size_t nomad_l1_scratch_bytes(void);
int nomad_l1_loss(nomad_ctx* ctx, const float* a_layers_dev, const float* b_layers_dev,
                  const float* a_emb_dev, const float* b_emb_dev, int B, int T,
                  float* loss_dev, void* scratch_dev, nomad_stream_t stream);

/* ---- training: Nomad.forward() as a differentiable loss (nomad.py:142-146 + autograd) ------ */
/*
 * The reference back-propagates through the whole backbone to `estimate` (and, wastefully, into the
 * backbone parameters, which nobody uses: the freeze is commented out at nomad.py:74-76).  Here the
 * weights are frozen and only d loss / d waveform is computed.
 *
 *   nomad_enable_backward    builds the transposed weight copies once and a 32 MB split-K scratch (allocates; call
 *                            before training).  With it, nomad_embed_train / nomad_embed_backward cut the contraction
 *                            of their small-M GEMMs (fewer than 512 tiles of 64 x 64, e.g. 32 clips of 1 s) into 2 or 4
 *                            fixed slices summed in order - deterministic, but a different rounding than nomad_embed's,
 *                            whose summation order never depends on the batch.  Not in fine-tuning mode.
 *   nomad_embed_train        = nomad_embed with layers_dev mandatory; additionally fills `saved_dev`
 *                              (nomad_saved_bytes) with what the backward needs
 *   nomad_l1_loss_backward   d NomadLoss / d a_layers, d a_emb  (sign(a-b)/numel, times *upstream_dev)
 *   nomad_embed_backward     given d loss / d layers [12][B][T][768] (nullable) and d loss / d emb [B][256]
 *                            writes d loss / d wav [B][n_samples]; scratch: nomad_backward_workspace_bytes;
 *                            the gradient entering the conv feature extractor is scaled by feature_grad_mult (below)
 */
int nomad_enable_backward(nomad_ctx* ctx);
/*
 * fairseq's Wav2Vec2Model wraps the conv feature extractor's output in GradMultiply(features, feature_grad_mult)
 * whenever feature_grad_mult != 1 (wav2vec2.py, Wav2Vec2Model.forward; also in eval mode): the forward is the
 * identity, the gradient flowing back INTO the extractor - and so d loss / d waveform of Nomad.forward(),
 * nomad.py:142-146 - is multiplied by it.  The wav2vec 2.0 BASE config that wav2vec_small.pt carries (and that
 * load_model_ensemble_and_task keeps, nomad.py:58) has feature_grad_mult = 0.1, which is the default here;
 * 1.0 gives the plain chain rule, 0 means "extractor under no_grad" in fairseq and yields dwav = 0.
 * Applies to nomad_embed_backward only (the fine-tuning step never reaches the frozen extractor).
 */
int nomad_set_feature_grad_mult(nomad_ctx* ctx, float mult);
/*
 * Arithmetic of the fp32-layout GEMMs (nomad_embed, nomad_embed_ragged, nomad_embed_train, nomad_embed_backward,
 * nomad_train_backward): mode 0 (default) exact fp32 MFMA (v_mfma_f32_16x16x4_f32, the reference's arithmetic); mode 1
 * "bf16x3 products": every product a*w as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on the bf16 matrix cores, hi = bf16(x),
 * lo = bf16(x - hi) made in registers, fp32 accumulation - all buffers, epilogues and every other kernel unchanged.  A GEMM
 * is then within ~3e-5 (relative) of the fp32 one and its K loop ~5x shorter: for the small-M problems of Nomad.forward()
 * (nomad.py:142-146 at the shapes of nomad_loss_test.py:60-79) and of the fine-tuning step.  Opt-in: Nomad(precision=
 * "bf16x3").  The pos-conv and attention kernels stay fp32.
 */
int nomad_set_gemm_precision(nomad_ctx* ctx, int mode);
int nomad_get_gemm_precision(const nomad_ctx* ctx, int* mode);
int nomad_get_feature_grad_mult(const nomad_ctx* ctx, float* mult);
int nomad_saved_bytes(const nomad_ctx* ctx, int B, int n_samples, size_t* bytes);
int nomad_backward_workspace_bytes(const nomad_ctx* ctx, int B, int n_samples, size_t* bytes);
int nomad_embed_train(nomad_ctx* ctx, const float* wav_dev, int B, int n_samples,
                      const float* head_w_dev, const float* head_b_dev, float* emb_dev, float* layers_dev,
                      void* saved_dev, size_t saved_bytes, void* workspace_dev, size_t workspace_bytes,
                      nomad_stream_t stream);
int nomad_l1_loss_backward(nomad_ctx* ctx, const float* a_layers_dev, const float* b_layers_dev,
                           const float* a_emb_dev, const float* b_emb_dev, int B, int T,
                           const float* upstream_dev, float* dlayers_dev, float* demb_dev, nomad_stream_t stream);
int nomad_embed_backward(nomad_ctx* ctx, const float* wav_dev, int B, int n_samples,
                         const float* head_w_dev, const float* head_b_dev, const float* layers_dev,
                         const void* saved_dev, size_t saved_bytes, const float* dlayers_dev, const float* demb_dev,
                         float* dwav_dev, void* workspace_dev, size_t workspace_bytes, nomad_stream_t stream);

/* ---- triplet fine-tuning step (src/training/train_triplet.py:112-133, src/config/train_triplet.yaml) ---- */
/*
 * The reference fine-tunes wav2vec 2.0 + head with A/P/N forwards, nn.TripletMarginLoss(margin),
 * loss.backward() and Adam (1e-5 on the backbone, `lr` on embedding_layer), conv feature extractor frozen
 * (freeze_convnet: True, the shipped config; False is supported too).  Here the trainable parameters live in ONE flat fp32 device vector (master copy in
 * checkpoint layout), with gradients and the two Adam moments in vectors of the same layout:
 *
 *   nomad_train_param_count   floats in the vector; head_begin = first float of embedding_layer (own lr)
 *   nomad_train_num_segments / nomad_train_segment   checkpoint key -> (offset, count) of the vector
 *   nomad_train_enable        allocates the vectors, fills the parameters from `host_weights` (the same struct
 *                             nomad_create took), re-points the engine at them (also calls nomad_enable_backward)
 *   nomad_train_zero_grad     gradients = 0
 *   nomad_embed_train         (above) forward that saves activations
 *   nomad_train_backward      given d loss / d emb [B][256]: ACCUMULATES d loss / d parameters into the gradient
 *                             vector (dW as MFMA GEMMs, split over the B*T contraction, fixed-order reduction);
 *                             scratch: nomad_train_workspace_bytes.  Stops at the feature extractor unless
 *                             nomad_train_set_convnet made it trainable.
 *   nomad_triplet_loss        loss [1] = mean_i max(||a-p+eps|| - ||a-n+eps|| + margin, 0), eps = 1e-6
 *                             (torch.pairwise_distance); da/dp/dn [B][256] nullable together (validation pass)
 *   nomad_train_adam_step     torch.optim.Adam update (amsgrad off, no weight decay) with the step count kept in
 *                             the context, then rebuilds the derived kernel-layout weights
 *   nomad_train_read / write  copy a whole vector out of / into the context (device pointers, async on stream);
 *                             what: 0 parameters, 1 gradients, 2 exp_avg, 3 exp_avg_sq
 *   nomad_train_set_step      set Adam's step counter (resume)
 *   nomad_train_set_stochastic   model.train() regularisation for the following nomad_embed_train /
 *                             nomad_train_backward calls: fairseq's dropout (after the encoder LayerNorm, out_proj
 *                             and fc2), attention_dropout (softmax probabilities), dropout_input (after
 *                             post_extract_proj) and LayerDrop (layer_mask bit l clear = layer l skipped).  Masks are a
 *                             counter-based hash of (seed, site, element): set the SAME values before a forward
 *                             and before its backward.  All zero + mask 0xFFF (the default) = eval-mode arithmetic.
 *   nomad_train_set_frozen    freeze_encoder != 0: the reference's `freeze_all: True` (train_triplet.py:76-79) - the conv feature
 *                             extractor AND the encoder (pos-conv, encoder LayerNorm, 12 layers) are frozen; gradients still
 *                             flow through them to post_extract_proj and the feature LayerNorm, which stay trainable with
 *                             the head.  Frozen parameters keep a zero gradient, so the Adam step leaves them untouched.
 *   nomad_train_set_convnet   trainable != 0: the reference's `freeze_convnet: False` (train_triplet.py:71-73) - the conv
 *                             feature extractor's weights and its GroupNorm get gradients too (conv dW as split-K MFMA GEMMs
 *                             over transposed im2col operands; scaled by feature_grad_mult like every gradient that
 *                             enters the extractor); changes nomad_train_workspace_bytes.  Default 0: their slices of the
 *                             gradient vector stay zero.
 *   nomad_train_set_branches  the batch of the following nomad_embed_train / nomad_train_backward calls is `branches`
 *                             equal groups of clips (anchor | positive | negative), each with its own LayerDrop mask -
 *                             as if each group had been its own forward call, but one launch sequence over all of
 *                             them wherever the masks agree.  branches = 1 (default): layer_mask of set_stochastic.
 */
int nomad_train_param_count(size_t* total, size_t* head_begin);
int nomad_train_num_segments(void);
int nomad_train_segment(int i, char* name, size_t name_cap, size_t* offset, size_t* count);
int nomad_train_enable(nomad_ctx* ctx, const nomad_weights* host_weights);
int nomad_train_workspace_bytes(const nomad_ctx* ctx, int B, int n_samples, size_t* bytes);
int nomad_train_zero_grad(nomad_ctx* ctx, nomad_stream_t stream);
int nomad_train_backward(nomad_ctx* ctx, const float* wav_dev, int B, int n_samples, const float* layers_dev,
                         const void* saved_dev, size_t saved_bytes, const float* demb_dev,
                         void* workspace_dev, size_t workspace_bytes, nomad_stream_t stream);
int nomad_triplet_loss(nomad_ctx* ctx, const float* a_dev, const float* p_dev, const float* n_dev, int B,
                       float margin, float* loss_dev, float* da_dev, float* dp_dev, float* dn_dev,
                       nomad_stream_t stream);
int nomad_train_adam_step(nomad_ctx* ctx, float lr_body, float lr_head, float beta1, float beta2, float eps,
                          nomad_stream_t stream);
int nomad_train_read(nomad_ctx* ctx, int what, float* dst_dev, nomad_stream_t stream);
int nomad_train_write(nomad_ctx* ctx, int what, const float* src_dev, nomad_stream_t stream);
int nomad_train_set_step(nomad_ctx* ctx, long long step);
int nomad_train_set_stochastic(nomad_ctx* ctx, float dropout, float attention_dropout, float dropout_input,
                               unsigned long long seed, unsigned layer_mask);
int nomad_train_set_branches(nomad_ctx* ctx, int branches, const unsigned* layer_masks);
int nomad_train_set_frozen(nomad_ctx* ctx, int freeze_encoder);
int nomad_train_set_convnet(nomad_ctx* ctx, int trainable);

/* ---- bf16 path (BASELINE config C5: long-form clips) ---------------------------------------- */
/*
 * The reference is fp32 only (torch 1.12, no AMP); this path exists for throughput on long clips.
 * Activations and weights are bf16 in HBM, every accumulation, bias/GELU/residual, GroupNorm/LayerNorm
 * statistic and the attention softmax are fp32.  Accuracy is reported against this library's fp32 path.
 *   nomad_enable_bf16          builds the bf16 weight copies once (allocates)
 *   nomad_embed_bf16           scoring forward: wav [B][n_samples] fp32 -> emb [B][256] fp32
 *   nomad_workspace_bytes_bf16 scratch size for it
 */
int nomad_enable_bf16(nomad_ctx* ctx);
int nomad_workspace_bytes_bf16(const nomad_ctx* ctx, int B, int n_samples, size_t* bytes);
int nomad_embed_bf16(nomad_ctx* ctx, const float* wav_dev, int B, int n_samples, float* emb_dev,
                     void* workspace_dev, size_t workspace_bytes, nomad_stream_t stream);
/* bf16 counterpart of nomad_embed_ragged (files of different lengths, e.g. long-form recordings through predict):
 * same arguments, no head override; every clip's result equals its own single-clip nomad_embed_bf16 call */
int nomad_workspace_bytes_ragged_bf16(const nomad_ctx* ctx, int B, const int* lengths_host, size_t* bytes);
int nomad_embed_ragged_bf16(nomad_ctx* ctx, const float* wav_dev, int B, int stride, const int* lengths_host,
                            float* emb_dev, void* workspace_dev, size_t workspace_bytes, nomad_stream_t stream);

/* ---- bf16x3: fp32-class scoring on the bf16 matrix cores -------------------------------------
 * Same surface as nomad_embed (TripletModel.forward, nomad.py:224-231), for callers who want the reference's
 * scores to its stated tolerance (1e-4) at more than the fp32 MFMA rate.  Every GEMM operand is kept as two bf16
 * planes, hi = bf16(x) and lo = bf16(x - hi) (16 mantissa bits, the bytes of one fp32), and multiplied as three
 * bf16 MFMA products hi*hi + hi*lo + lo*hi with fp32 accumulation - the conv stack, every dense layer, the grouped
 * pos-conv (as a Toeplitz GEMM over 5-frame blocks) and both attention products; bias, GELU, residuals, LayerNorm,
 * softmax and the head are fp32.  Accuracy is measured against the fp32 path in tests/test_gpu_bf16x3.py
 * (NOMAD scores agree to ~1e-6).  Forward only: no backward (the branch of nomad.forward() that carries the gradient
 * stays on nomad_embed_train).
 *   nomad_enable_bf16x3          builds the split weight copies (allocates once; call again after nomad_train_* /
 *                                weight updates)
 *   nomad_embed_bf16x3           wav [B][n_samples] fp32 -> emb [B][256] fp32
 *   nomad_workspace_bytes_bf16x3 scratch size for it
 */
int nomad_enable_bf16x3(nomad_ctx* ctx);
int nomad_workspace_bytes_bf16x3(const nomad_ctx* ctx, int B, int n_samples, size_t* bytes);
int nomad_embed_bf16x3(nomad_ctx* ctx, const float* wav_dev, int B, int n_samples, float* emb_dev,
                       void* workspace_dev, size_t workspace_bytes, nomad_stream_t stream);
/* LossNetLayers.forward (nomad.py:243-258) on the bf16x3 path - what the no-gradient `clean` branch of nomad.forward()
 * runs on: like nomad_embed with layers_out, i.e. optional head override (both or neither), emb [B][256] and
 * layers_out [12][B*T][768] fp32 (the fp32 LayerNorm output of every encoder layer).  Same workspace as
 * nomad_embed_bf16x3. */
int nomad_embed_layers_bf16x3(nomad_ctx* ctx, const float* wav_dev, int B, int n_samples, const float* head_w_dev,
                              const float* head_b_dev, float* emb_dev, float* layers_out_dev, void* workspace_dev,
                              size_t workspace_bytes, nomad_stream_t stream);
/* bf16x3 counterpart of nomad_embed_ragged (files of different lengths in one launch sequence - what predict uses):
 * same arguments, no head override; every clip's result equals its own single-clip nomad_embed_bf16x3 call */
int nomad_workspace_bytes_ragged_bf16x3(const nomad_ctx* ctx, int B, const int* lengths_host, size_t* bytes);
int nomad_embed_ragged_bf16x3(nomad_ctx* ctx, const float* wav_dev, int B, int stride, const int* lengths_host,
                              float* emb_dev, void* workspace_dev, size_t workspace_bytes, nomad_stream_t stream);

/* ---- WAV front end of the file-scoring loop (host only, no GPU work) ------------------------
 * Nomad.get_embeddings_csv (nomad.py:171-183) calls load_processing (nomad.py:192-212) per file:
 * torchaudio.load -> fp32 in [-1, 1), mean of the first two channels, resample to 16 kHz.  These two
 * entry points do the first two steps on plain host threads, straight into the rows of the (pinned)
 * staging buffer nomad_embed_ragged* reads after one H2D copy; nomad_wav_read_rows also resamples.
 * Decoded: PCM 8/16/24/32 (x 2^-(bits-1)), IEEE float 32/64, WAVE_FORMAT_EXTENSIBLE of those. */
typedef struct nomad_wav_info {
    int sample_rate, channels, format_tag, bits;
    long long frames;      /* per channel; a data chunk cut short by the end of the file counts what is there */
    long long data_offset; /* byte offset of the sample data */
} nomad_wav_info;
/* Headers of n files on `threads` host threads.  status[i] = NOMAD_OK, NOMAD_ERR_IO or NOMAD_ERR_FORMAT per
 * file (info[i] zeroed then); the return value is NOMAD_OK unless the arguments are bad. */
int nomad_wav_probe(const char* const* paths, int n, nomad_wav_info* info, int* status, int threads);
/* Frames a probed file has at target_rate: its own count, or ceil(frames * target / rate) after resampling. */
int nomad_wav_frames_at(const nomad_wav_info* info, int target_rate, long long* frames);
/* Sample data of n probed files -> dst[row[i] * stride + 0 .. frames_at(info[i], target_rate)) as mono fp32 at target_rate
 * (row == NULL: row[i] = i); the rest of a row is left untouched.  A file at another rate is resampled like
 * torchaudio.transforms.Resample(rate, target_rate) with its defaults (Hann-windowed sinc, lowpass_filter_width 6, rolloff
 * 0.99; nomad.py:203-205).  Needs frames_at <= stride.  status as above; returns the first non-zero status (all files are
 * attempted). */
int nomad_wav_read_rows(const char* const* paths, const nomad_wav_info* info, int n, const int* row,
                        float* dst_host, long long stride, int target_rate, int* status, int threads);

/* ---- measurement ------------------------------------------------------------------------- */
/* Kernel classes for the in-library HIP-event timers. */
enum {
    NOMAD_K_GEMM = 0,       /* every GEMM launch (all instantiations) */
    NOMAD_K_ATTN = 1,
    NOMAD_K_FRONT = 2,
    NOMAD_K_ROW = 3,
    NOMAD_K_PAIR = 4,
    NOMAD_K_GEMM_BIG = 5,   /* of which: the 256x128 instantiation (gemm_*_glds_kernel<256,128,...>; bf16: 128x128 / 256x256) */
    NOMAD_K_GEMM_FINE = 6,  /* of which: the finer instantiations (128x128x32, 128x64x32, N = 48) */
    NOMAD_K_COUNT = 7
};
/* When enabled every kernel launch of nomad_embed/nomad_pairwise is bracketed by hipEvents on the
 * launch stream.  nomad_profile_read synchronises those events and returns, per class, the summed
 * device time (ms), launch count and algorithmic FLOPs (2*M*N*K of the true problem, no padding)
 * since the last nomad_profile_reset. */
int nomad_profile_enable(nomad_ctx* ctx, int on);
int nomad_profile_reset(nomad_ctx* ctx);
int nomad_profile_read(nomad_ctx* ctx, double ms[NOMAD_K_COUNT], long long launches[NOMAD_K_COUNT],
                       double flops[NOMAD_K_COUNT]);

/* ---- kernel-level diagnostics (used by tests/ to localise a parity failure) --------------- */
/* C[M][N] = epilogue(A[M][K] * W[N][K]^T): + bias[N] (nullable), GELU if gelu!=0, + R[M][N] (nullable).
 * tile selects the kernel instantiation.  libnomad_hip.so holds the ones the forward / backward select: 33 = 256x128x16
 * (3-stage LDS-DMA), 31 = 128x128x32, 20 = 128x128x32 (4 waves), 34 = 128x64x32, 37 = 64x64x32, 48 = N = 48 (pos-conv).
 * Every other id (register-staged kernels, ablations, A/B variants) exists in libnomad_diag.so only - the same source
 * built with -DNOMAD_DIAG for the measurement tools and kernel tests - and returns NOMAD_ERR_INVALID here. */
int nomad_diag_gemm(nomad_ctx* ctx, const float* A_dev, const float* W_dev, const float* bias_dev,
                    const float* R_dev, float* C_dev, int M, int N, int K, int gelu, int tile,
                    nomad_stream_t stream);
/* The bf16 GEMM: A [M][K], W [N][K], R, C [M][N] are bf16; bias fp32.  tile (libnomad_hip.so): 1 = 128x128,
 * 2 = 128x64, 3 = 256x256, 4 = 64x64 (all BK = 64), 16 = 256x256 deep-pipelined; others: libnomad_diag.so. */
int nomad_diag_gemm_bf16(nomad_ctx* ctx, const void* A_dev, const void* W_dev, const float* bias_dev,
                         const void* R_dev, void* C_dev, int M, int N, int K, int gelu, int tile,
                         nomad_stream_t stream);
/* bf16x3 pieces.  nomad_diag_split_bf16: fp32 in[n] -> split planes out (hi at 0, lo at `plane` bf16 elements), or back
 * (inverse != 0: `in` is the split buffer, `out` fp32).  nomad_diag_gemm_bf16x3: the split GEMM on split A [M][K]
 * (planes M*K apart), W [N][K] (N*K apart), optional R [M][N] (M*N apart); C is split (M*N apart) or fp32 (out_f32). */
int nomad_diag_split_bf16(nomad_ctx* ctx, const float* in_dev, void* out_dev, long long plane, long long n, int inverse,
                          nomad_stream_t stream);
int nomad_diag_gemm_bf16x3(nomad_ctx* ctx, const void* A_dev, const void* W_dev, const float* bias_dev,
                           const void* R_dev, void* C_dev, int M, int N, int K, int gelu, int out_f32,
                           nomad_stream_t stream);
/* bf16x3 attention: split qkv [B*T][2304] (planes B*T*2304 apart, q pre-scaled) -> split out [B*T][768].
 * waves: -1 = what the forward uses, 0 = the tiled kernel, 4 / 8 = the K/V-resident kernel (T <= 256) with that many
 * waves per (clip, head). */
int nomad_diag_attention_bf16x3(nomad_ctx* ctx, const void* qkv_dev, void* out_dev, int B, int T, int waves,
                                nomad_stream_t stream);
/* bf16 attention: qkv [B*T][2304] bf16 (q pre-scaled by 64^-0.5) -> out [B*T][768] bf16.  q_has_log2e != 0: q is
 * additionally scaled by log2(e), as the bf16 forward's QKV projection produces it (the kernel works in log2 units);
 * 0: plain q, scaled (and re-rounded to bf16) inside the kernel. */
int nomad_diag_attention_bf16(nomad_ctx* ctx, const void* qkv_dev, void* out_dev, int B, int T, int q_has_log2e,
                              nomad_stream_t stream);
/* out[M][N] = LayerNorm(in[M][N]) * gamma + beta, N in {512, 768}, eps 1e-5. */
/* one wave spins for spin_ticks of the 100 MHz wall counter; out_dev[0] = shader-clock cycles elapsed, out_dev[1] = wall
 * ticks: run it on a second stream to read the clock a kernel under test actually gets */
int nomad_diag_clock_probe(nomad_ctx* ctx, unsigned long long spin_ticks, unsigned long long* out_dev,
                           nomad_stream_t stream);
int nomad_diag_layernorm(nomad_ctx* ctx, const float* in_dev, const float* gamma_dev, const float* beta_dev,
                         float* out_dev, int M, int N, nomad_stream_t stream);
/* ctx_out[B*T][768] = softmax(q k^T) v per head, from qkv[B*T][2304] (q pre-scaled). */
int nomad_diag_attention(nomad_ctx* ctx, const float* qkv_dev, float* out_dev, int B, int T, nomad_stream_t stream);
/* dx[M][N] = LayerNorm backward of g w.r.t. the LN input x (weights frozen). */
int nomad_diag_layernorm_bwd(nomad_ctx* ctx, const float* x_dev, const float* g_dev, const float* gamma_dev,
                             float* dx_dev, int M, int N, nomad_stream_t stream);
/* Runs the attention forward (ctx_out [B*T][768], lse [B*12][T]) then its backward: dqkv [B*T][2304]. */
int nomad_diag_attention_bwd(nomad_ctx* ctx, const float* qkv_dev, const float* dctx_dev, float* ctx_out_dev,
                             float* lse_dev, float* dqkv_dev, int B, int T, nomad_stream_t stream);
/* With on!=0 every conv layer output gets its own workspace region (no ping-pong aliasing), so
 * nomad_diag_workspace_region can read all of them back after nomad_embed.  Changes
 * nomad_workspace_bytes. */
int nomad_diag_keep_intermediates(nomad_ctx* ctx, int on);
/* Byte offset and size of a named intermediate inside the nomad_embed workspace for (B, n_samples):
 * "conv0".."conv6" (time-major [B][L_i][512]), "featln" [B*T][512], "xpad" [16 groups][B][T+128][48]
 * (post_extract_proj output, group-major, at frames 64..64+T of each clip; zero frames around). */
int nomad_diag_workspace_region(const nomad_ctx* ctx, int B, int n_samples, const char* name,
                                size_t* offset, size_t* bytes);

#ifdef __cplusplus
}
#endif
#endif /* NOMAD_HIP_H */
