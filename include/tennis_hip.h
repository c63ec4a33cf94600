/*
 * tennis_hip.h — C ABI of libtennis_hip.so: the MI355X (gfx950) hot path of
 * HaydenFaulkner/Tennis, hand-written HIP behind the reference's model surface.
 *
 * The reference has NO FFI boundary of its own (pure Python on MXNet,
 * SURVEY §8b).  Each entry point below replaces the device work that one
 * reference call performs inside MXNet; the cited file:line is the reference
 * call site that a maintainer would re-point at this library (INTEGRATION.md
 * shows the ctypes stub).
 *
 * Conventions
 *   - every function returns 0 on success, a negative tn_status otherwise;
 *     tn_last_error() gives a thread-local message for the last failure;
 *   - all tensor pointers are DEVICE pointers on the ctx's device unless the
 *     parameter name ends in _host; the caller owns every input/output buffer;
 *     the library owns weights and workspace inside opaque handles;
 *   - calls on one tn_ctx are ordered on its HIP stream and asynchronous;
 *     tn_ctx_sync() waits.  A ctx is not re-entrant; use one ctx per thread.
 *   - no allocation happens in *_forward.
 */
#ifndef TENNIS_HIP_H
#define TENNIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  TN_OK = 0,
  TN_ERR_INVALID = -1,   /* bad argument / shape (mirrors the reference's asserts) */
  TN_ERR_HIP = -2,       /* HIP runtime error */
  TN_ERR_MISSING = -3,   /* a required parameter name was not supplied */
  TN_ERR_NOMEM = -4
} tn_status;

/* Input frame layouts accepted by tn_densenet121_forward. */
typedef enum {
  TN_LAYOUT_NCHW_F32 = 0, /* reference layout: ToTensor+Normalize output (evaluate.py:96-97) */
  TN_LAYOUT_NHWC_F16 = 1, /* native: normalised frames, fp16, 3 channels */
  TN_LAYOUT_NHWC_U8 = 2   /* decoded RGB bytes; ToTensor+Normalize fused into the stem load */
} tn_layout;

typedef enum { TN_RNN_GRU = 0, TN_RNN_LSTM = 1 } tn_rnn_kind;
typedef enum { TN_POOL_MAX = 0, TN_POOL_MEAN = 1 } tn_pool_kind;

/* A named host tensor (Gluon parameter name -> fp32 data). */
typedef struct {
  const char *name;
  const float *data_host;
  int64_t numel;
} tn_param;

typedef struct tn_ctx tn_ctx;
typedef struct tn_encoder tn_encoder;
typedef struct tn_dense tn_dense;
typedef struct tn_birnn tn_birnn;
typedef struct tn_gnmt tn_gnmt;

int tn_version(void);
const char *tn_last_error(void);

/* ---- context: one per (device, stream) --------------------------------- */
/* `stream` is the hipStream_t to launch on (e.g. torch's current stream; NULL is
 * HIP's default stream).  With own_stream != 0 the library creates and owns a
 * non-blocking stream instead and `stream` is ignored.  Replaces: mx.gpu(i)
 * context selection, reference evaluate.py:85. */
int tn_ctx_create(int device, void *stream, int own_stream, tn_ctx **out);
void *tn_ctx_stream(tn_ctx *ctx);
int tn_ctx_sync(tn_ctx *ctx);       /* replaces the implicit sync of .asnumpy(), evaluate.py:314 */
int tn_ctx_destroy(tn_ctx *ctx);

/* ---- frame encoder: DenseNet-121 .features ------------------------------ */
/* Replaces get_model('DenseNet121', pretrained=True).features (reference
 * evaluate.py:125, train.py:204, train_gnmt.py:150) and its forward
 * net.backbone(x) (evaluate.py:313).  `params` are Gluon-named fp32 host
 * tensors "<prefix>conv0_weight", "<prefix>stage1_batchnorm0_gamma", ...
 * (H,W) is the input size (224 or 512 in the reference), feature_dim is
 * 1024*floor(H/32/7)*floor(W/32/7) (1024 @224, 4096 @512; train.py:259). */
int tn_densenet121_create(tn_ctx *ctx, const tn_param *params, int n_params, const char *prefix,
                          int height, int width, int max_batch, tn_encoder **out);
/* The same with flags.  TN_ENC_EXACT_WEIGHTS: the fp32 convolution weights of the dense layers and transitions are
 * kept as two fp16 numbers each (hi + lo, 22 bits of the fp32 weight) and every product is formed with both: features
 * within 1e-3 of the fp32 reference evaluated on the UN-rounded weights (reference models/vision/definitions.py:27-33
 * evaluates fp32 parameters), at twice the MFMA work of those layers.  Without the flag the weights are rounded to
 * fp16 once (the served fp16 model).  Any input size since round 6 (maps no fused kernel tiles run their layers un-fused with the
 * same hi + lo pass; flat frames at 448 x 448 / 512 x 512 keep a tail of 1.2e-3 that is the fp16 activation path's: DESIGN.md section 4). */
#define TN_ENC_EXACT_WEIGHTS 1
int tn_densenet121_create_ex(tn_ctx *ctx, const tn_param *params, int n_params, const char *prefix, int height, int width,
                             int max_batch, int flags, tn_encoder **out);
int tn_densenet121_feature_dim(const tn_encoder *enc);
size_t tn_densenet121_workspace_bytes(const tn_encoder *enc);
/* x: `batch` frames in `layout`; feat: (batch, feature_dim) fp32, NCHW-flatten order. */
int tn_densenet121_forward(tn_encoder *enc, const void *x, tn_layout layout, int batch, float *feat);
/* Pipelined forwards.  A large batch runs as two half batches on two library-owned streams; by default forward makes
 * the ctx's stream wait for both before it returns, so the call is stream-ordered like every other one.  With
 * set_pipelined(enc, 1) forward does NOT: consecutive forwards then overlap (the first half of call i+1 starts beside
 * the tail of the second half of call i, which runs on half of the CUs), and the CALLER orders the results:
 * tn_densenet121_join(enc, 0) makes the ctx's stream wait for the last forward, join(enc, 1) for the one before it
 * (results consumed one call behind).  Until its join, a forward's input x must not be overwritten and its feat not
 * read; set_pipelined(enc, 0) joins everything outstanding.  Replaces nothing in the reference (MXNet's engine
 * overlaps independent ops by itself); it exists for streaming a corpus through the encoder (config C4).
 * (Round 6: from 128 frames per call on, a pipelined forward runs its WHOLE batch on one of the two streams, consecutive calls
 * alternating, each in its own workspace set - at most two forwards are in flight, as before; same results bit for bit,
 * same join contract; TN_NO_INTERLEAVE=1 in the environment keeps the half-batch form.) */
int tn_densenet121_set_pipelined(tn_encoder *enc, int on);
int tn_densenet121_join(tn_encoder *enc, int lag);
/* Calibration statistics for the calibrated fp16 conversion (tennis_amd/weights.py::as_fp16_model(input_means=...)):
 * `batch` frames go through the layer-wise kernels, and for each of the 119 convolutions behind the stem, in execution order
 * (per dense layer the 1x1's K input channels, then the 3x3's 128; a transition's inputs after its block), the mean over all
 * pixels of the OPERAND that convolution's folded weights multiply is written to means_host (fp32, *numel values): for a dense
 * layer's 1x1 the clamped stored activation clamp(x, lo, hi) (tn_bn_relu_clamp_fold; relu(bn(x)) = sw * that + tc), for its 3x3
 * the ReLU'd bottleneck, for a transition relu(bn(x)).  (The stem's operand mean, x - 255 mean_c, is a property of the frame and
 * is computed by the caller.) */
int tn_densenet121_input_means(tn_encoder *enc, const void *x, tn_layout layout, int batch, float *means_host,
                               int64_t capacity, int64_t *numel);
/* Host side of that conversion (no GPU involved; csrc/calib_host.hip): w (N, K) fp32 -> out (N, K), every entry one of the two
 * fp16 neighbours of w's entry, chosen per output row so that the row's rounding error is (nearly) orthogonal to all F rows of
 * A (F, K) - the per-frame mean input activations of F calibration frames (one tn_densenet121_input_means call per frame) -
 * by a greedy pass and `sweeps` passes of coordinate descent on || A d ||^2 / F + ridge * sum_k (d[k] rms_f A[f][k])^2.
 * The reference evaluates fp32 parameters (models/vision/definitions.py:27-33): this keeps ONE fp16 number per weight within
 * the 1e-3 bar of that evaluation. */
int tn_round_fp16_calibrated(const float *w, int rows, int cols, const double *A, int frames, int sweeps, double ridge, float *out);
/* The `.npy` side of `evaluate.py --save_feats` (reference evaluate.py:306-321: np.save(feat_path, feat[i]) per frame unless the file
 * exists): a pool of host threads that creates the directories and writes one NumPy-format-1.0 float32 file per row, byte for byte
 * what np.save writes.  submit copies the rows (host memory) and returns; drain waits for everything submitted, returns the totals
 * since creation and fails with the first file error.  Host code, no GPU involved (csrc/npy_host.hip). */
typedef struct tn_npy_writer tn_npy_writer;
int tn_npy_writer_create(int threads, tn_npy_writer **out);
int tn_npy_writer_submit(tn_npy_writer *w, const float *rows_host, int n, int dim, const char *const *paths, int skip_existing);
int tn_npy_writer_drain(tn_npy_writer *w, int64_t *written, int64_t *skipped);
int tn_npy_writer_destroy(tn_npy_writer *w);
/* BatchNorm -> ReLU in front of a dense layer's 1x1 convolution (gluoncv DenseNet _make_dense_layer, reference call site
 * models/vision/definitions.py:30) in the fused kernels' rounding-free form: relu(scale x + shift) = sw clamp(x, lo, hi) + tc with
 * lo / hi fp16 numbers (the ReLU threshold -shift / scale on the side the scale's sign says, +-65504 on the other); sw[k]
 * multiplies column k of the 1x1 weights before they are rounded to fp16 - part of how the fp16 model is defined
 * (weights.as_fp16_model calls this so that the host and the library agree bit for bit) - and sum_k w[n][k] tc[k] joins the shift
 * of the BatchNorm behind the convolution.  Host code, no GPU involved (csrc/calib_host.hip). */
int tn_bn_relu_clamp_fold(const float *gamma, const float *beta, const float *running_mean, const float *running_var, int channels,
                          float *lo, float *hi, float *sw, float *tc);
int tn_densenet121_destroy(tn_encoder *enc);

/* ---- nn.Dense(units, flatten=True) -------------------------------------- */
/* Replaces FrameModel.classes / CNNRNN.classes (reference
 * models/vision/definitions.py:25,32,101,108-109).  y = x W^T + b, fp32. */
int tn_dense_create(tn_ctx *ctx, const float *weight_host, const float *bias_host, int units,
                    int in_units, tn_dense **out);
int tn_dense_forward(tn_dense *d, const float *x, int rows, float *y);
int tn_dense_destroy(tn_dense *d);

/* ---- gluon.rnn.GRU/LSTM(hidden, layout='NTC', bidirectional=True) ------- */
/* Replaces CNNRNN.rnn (reference models/vision/definitions.py:94-96,106) and
 * one bidirectional layer of GNMTEncoder (models/captioning/gnmt.py:141-148).
 * params: "<prefix>{l,r}0_{i2h,h2h}_{weight,bias}"; gate order GRU [r,z,n],
 * LSTM [i,f,g,o].  x (B,T,F) fp32 -> seq (B,T,2H) fp32 = concat(fwd,bwd);
 * zero initial state.  valid_len (B,) int32 or NULL: steps >= valid_len are
 * skipped and emit zeros; the reverse direction starts at valid_len-1.
 * h_last/c_last (2,B,H) fp32 or NULL receive the final states [fwd,bwd]. */
int tn_birnn_create(tn_ctx *ctx, tn_rnn_kind kind, int input_size, int hidden, const tn_param *params,
                    int n_params, const char *prefix, int bidirectional, int max_rows, tn_birnn **out);
int tn_birnn_forward(tn_birnn *r, const float *x, int batch, int steps, const int32_t *valid_len,
                     float *seq, float *h_last, float *c_last);
int tn_birnn_destroy(tn_birnn *r);

/* ---- test transform, geometric half: Resize + CenterCrop on decoded frames ----- */
/* Replaces transforms.Resize(data_shape + 32) + transforms.CenterCrop(data_shape) of the reference's
 * transform_test (evaluate.py:93-96; gluon Resize -> mx.image.imresize interp=1 = cv::resize INTER_LINEAR on
 * 8-bit RGB [EXT]; CenterCrop offset int((s - c) / 2)).  src (batch, src_h, src_w, 3) uint8 RGB, dst
 * (batch, crop, crop, 3) uint8 = TN_LAYOUT_NHWC_U8 for tn_densenet121_forward, whose stem load applies
 * ToTensor + Normalize (evaluate.py:96-97).  Both DEVICE buffers; bit-exact integer arithmetic. */
typedef struct tn_preproc tn_preproc;
int tn_preproc_create(tn_ctx *ctx, int src_h, int src_w, int resize, int crop, tn_preproc **out);
int tn_preproc_forward(tn_preproc *p, const uint8_t *src, int batch, uint8_t *dst);
int tn_preproc_destroy(tn_preproc *p);
/* Train-time augmentation of the frame classifier (reference train.py:127-136: transforms.RandomResizedCrop(data_shape),
 * RandomFlipLeftRight(), RandomColorJitter(brightness, contrast, saturation), RandomLighting(alpha) in front of ToTensor / Normalize):
 * the image arithmetic for a batch, one parameter record per frame.  The random draws are the caller's (MXNet's generator cannot be
 * reproduced: tennis_amd/transforms.py draws them the way mx.image.random_size_crop / the image_random operators do).  src (batch,
 * src_h, src_w, 3) uint8 RGB and every other buffer on the device, frames_host = the same records on the host (validated there);
 * tmp batch * size * size * 3 bytes, gray_tmp batch floats; dst (batch, size, size, 3) uint8 - what ToTensor receives. */
typedef struct tn_aug_frame {
  int32_t x0, y0, cw, ch;        /* crop window (random_size_crop), resized to size x size with cv::resize(INTER_LINEAR) */
  int32_t flip;                  /* RandomFlipLeftRight */
  int32_t order;                 /* four 2-bit fields, first applied in the low bits: 0 brightness, 1 contrast, 2 saturation, 3 hue (= nothing) */
  float brightness, contrast, saturation;   /* the operators' alphas: 1 + U(-p, p) */
  float light[3];                /* RandomLighting: eigvec (alpha * eigval), per channel */
} tn_aug_frame;
int tn_augment_forward(tn_ctx *ctx, const uint8_t *src, int batch, int src_h, int src_w, const tn_aug_frame *frames_dev,
                       const tn_aug_frame *frames_host, int size, uint8_t *tmp, float *gray_tmp, uint8_t *dst);
/* ToTensor + Normalize (reference evaluate.py:96-97, train.py:138-139) for consumers of fp32 frames (the fine-tuning
 * step; the inference encoder fuses it into its stem load): dst[p][c] = (src[p][c] / 255 - mean[c]) / std[c],
 * src (pixels, 3) uint8 NHWC, dst (pixels, 3) float NHWC, both DEVICE; mean3 / std3 HOST arrays of 3 floats. */
int tn_to_tensor_normalize(tn_ctx *ctx, const uint8_t *src, long pixels, const float *mean3, const float *std3, float *dst);

/* ---- JPEG decode on the device ------------------------------------------------ */
/* Replaces mx.image.imread(path, 1) of the reference's frame loader (dataset.py:204,216: OpenCV imdecode -> libjpeg
 * with its default parameters, JDCT_ISLOW and fancy upsampling) for a batch of files of ONE geometry (the frames of
 * a video): baseline / extended sequential Huffman JPEG (SOF0, SOF1), 8 bit, one interleaved scan, grey or YCbCr
 * 4:4:4 / 4:2:2 / 4:2:0, restart intervals included.  data_host[i] / sizes[i]: the i-th file's bytes in HOST memory;
 * rgb: DEVICE buffer (n, height, width, 3) uint8 RGB = the src of tn_preproc_forward (grey files are replicated to
 * three channels, as imread flag=1 does).  Only the marker segments are read on the host; Huffman decoding
 * (parallel inside each file), IDCT, upsampling and colour conversion run on the device, bit-exact with libjpeg's
 * integer arithmetic.  Anything else (progressive, arithmetic coding, 12 bit, CMYK, multi-scan, files of different
 * geometry in one call, corrupt data) is refused with TN_ERR_INVALID and a message naming the file.  The call
 * synchronises the ctx's stream (it reads back the convergence flag of the parallel Huffman decoder) and may grow
 * the handle's workspace.  tn_jpeg_info parses one file's header on the host (no device work). */
typedef struct tn_jpeg tn_jpeg;
int tn_jpeg_create(tn_ctx *ctx, tn_jpeg **out);
int tn_jpeg_info(const uint8_t *data_host, size_t size, int *width, int *height, int *components, int *h_samp, int *v_samp);
int tn_jpeg_decode(tn_jpeg *j, const uint8_t *const *data_host, const size_t *sizes, int n, uint8_t *rgb, int *width, int *height);
int tn_jpeg_destroy(tn_jpeg *j);

/* ---- F.max / F.mean over axis 1 ------------------------------------------ */
/* Replaces reference models/vision/definitions.py:66-69,107.  x (B,T,F) -> y (B,F). */
int tn_temporal_pool(tn_ctx *ctx, const float *x, int batch, int steps, int feat, tn_pool_kind kind,
                     float *y);

/* ---- temporal-head training step ------------------------------ */
/* bi-GRU / bi-LSTM(hidden) (`kind`; CNNRNN type='gru'|'lstm', definitions.py:93-96; LSTM gates [i,f,g,o]) over
 * features -> max over T -> Dense(classes) -> SoftmaxCrossEntropyLoss, backward, SGD with
 * momentum and weight decay: the frozen-backbone recipe of reference train.py (gluon.Trainer(..., 'sgd', {lr,
 * momentum, wd}) :298-299; SoftmaxCrossEntropyLoss :324; ag.record / ag.backward / trainer.step(batch_size)
 * :410-424) on the CNNRNN model of models/vision/definitions.py:94-110 in feature mode.
 * Parameters use the Gluon names (<rnn_prefix>{l0,r0}_{i2h,h2h}_{weight,bias}, <dense_prefix>{weight,bias}).
 * forward_backward leaves d(sum of per-sample losses)/d(param) in the gradient buffer; the caller may all-reduce
 * that buffer over ranks (tn_head_buffers gives the flat device arrays) before tn_head_sgd_step, whose update is
 * MXNet's sgd_mom_update: mom = momentum*mom - lr*(rescale_grad*grad + wd*w); w += mom. */
typedef struct tn_head tn_head;
int tn_head_create(tn_ctx *ctx, tn_rnn_kind kind, int input_size, int hidden, int classes, const tn_param *params,
                   int n_params, const char *rnn_prefix, const char *dense_prefix, int max_batch, int max_steps, tn_head **out);
int tn_head_forward_backward(tn_head *h, const float *x, const int32_t *labels, int batch, int steps, float *loss,
                             float *logits);
int tn_head_buffers(tn_head *h, float **params_dev, float **grads_dev, int64_t *numel);
int tn_head_sgd_step(tn_head *h, float lr, float momentum, float wd, float rescale_grad);
int tn_head_read_param(tn_head *h, const char *name, int gradient, float *out_host, int64_t capacity, int64_t *numel);
int tn_head_destroy(tn_head *h);

/* ---- fine-tuning step of the frame classifier (second half) ------ */
/* FrameModel(DenseNet-121 .features, Dense(classes)) trained end to end as reference train.py does when the backbone is
 * not frozen: BatchNorm in training mode (batch statistics; running statistics updated with momentum 0.9),
 * SoftmaxCrossEntropyLoss per sample (train.py:324), backward of the summed losses (:419-421), SGD with momentum and
 * weight decay, rescale_grad = 1/batch_size (:298-299,424).  fp32.  Parameters by their Gluon names (as
 * tn_densenet121_create + the Dense).  x (batch, H, W, 3) fp32 normalised NHWC frames and labels (batch,) int32 are
 * DEVICE buffers; batch, height and width must equal the handle's (the frame buffer is read as batch x height x width x 3).  read_param returns Gluon-ordered weights / gradients, running
 * statistics, or "<bn>_batch_mean" / "<bn>_batch_var" of the last step. */
typedef struct tn_finetune tn_finetune;
int tn_finetune_create(tn_ctx *ctx, const tn_param *params, int n_params, const char *backbone_prefix,
                       const char *dense_prefix, int height, int width, int classes, int batch, tn_finetune **out);
int tn_finetune_forward_backward(tn_finetune *f, const float *x, const int32_t *labels, int batch, int height, int width, float *loss,
                                 float *logits);
int tn_finetune_buffers(tn_finetune *f, float **params_dev, float **grads_dev, int64_t *numel);
int tn_finetune_sgd_step(tn_finetune *f, float lr, float momentum, float wd, float rescale_grad);
int tn_finetune_read_param(tn_finetune *f, const char *name, int gradient, float *out_host, int64_t capacity,
                           int64_t *numel);
int tn_finetune_destroy(tn_finetune *f);

/* ---- captioner training step ---------------------------------- */
/* One step of reference train_gnmt.py::train (:328-337) for GRU or LSTM cells (--cell_type), num_layers = 2,
 * num_bi_layers = 1:
 *   out, _ = model(src, tgt[:, :-1], src_valid_length, tgt_valid_length - 1)        teacher-forced NMTModel.forward
 *   loss = MaskedSoftmaxCELoss(out, tgt[:, 1:], tgt_valid_length - 1).mean()
 *          * (tgt.shape[1] - 1) / (tgt_valid_length - 1).mean()                      = summed NLL / number of valid tokens
 *   loss.backward(); gluon.Trainer(params, 'adam', {'learning_rate': lr}).step(1)   MXNet Adam [EXT]
 * Parameters use the names of tn_gnmt_create.  forward_backward takes DEVICE buffers: src (batch, steps, input_size)
 * fp32, tgt (batch, ld) int32 of which tgt_len columns are used, valid lengths (batch,) int32 (tgt_valid_len counts BOS
 * and EOS, as the reference's data loader gives it); loss is one device float, logits_out (batch, tgt_len-1, vocab)
 * optional.  The gradient of the loss w.r.t. every parameter is left in the flat gradient buffer
 * (tn_gnmt_trainer_buffers; all-reduce it over ranks for data parallelism) for tn_gnmt_trainer_adam_step. */
typedef struct tn_gnmt_trainer tn_gnmt_trainer;
int tn_gnmt_trainer_create(tn_ctx *ctx, const tn_param *params, int n_params, const char *prefix, tn_rnn_kind cell_kind,
                           int input_size, int hidden, int embed, int vocab, int max_batch, int max_src_len,
                           int max_tgt_len, tn_gnmt_trainer **out);
/* Any layer count the inference handle serves (tn_gnmt_create_ex): num_layers >= 2, 0 <= num_bi_layers < num_layers, flags =
 * TN_GNMT_USE_RESIDUAL or 0 - the arguments the reference passes into the model it trains (train_gnmt.py:58-61,223-227;
 * models/captioning/gnmt.py:71-111,153-157,393-396).  Parameter names: enc_rnn{i}_l_ / _r_ (bidirectional layers), enc_rnn{i}_,
 * dec_rnn{j}_.  tn_gnmt_trainer_create = (2, 1, 0), the reference's flag defaults. */
int tn_gnmt_trainer_create_ex(tn_ctx *ctx, const tn_param *params, int n_params, const char *prefix, tn_rnn_kind cell_kind,
                              int input_size, int hidden, int embed, int vocab, int num_layers, int num_bi_layers, int flags,
                              int max_batch, int max_src_len, int max_tgt_len, tn_gnmt_trainer **out);
int tn_gnmt_trainer_forward_backward(tn_gnmt_trainer *t, const float *src, const int32_t *src_valid_len,
                                     const int32_t *tgt, int ld, const int32_t *tgt_valid_len, int batch, int steps,
                                     int tgt_len, float *loss, float *logits_out);
int tn_gnmt_trainer_buffers(tn_gnmt_trainer *t, float **params_dev, float **grads_dev, int64_t *numel);
/* --dropout of train_gnmt.py (gnmt.py:152,395: after each encoder layer and on the top decoder cell's output), inverted
 * dropout from a counter-based generator; 0 until set.  tn_gnmt_trainer_dropout_masks: the last step's masks (test hook). */
int tn_gnmt_trainer_set_dropout(tn_gnmt_trainer *t, float p, uint64_t seed);
int tn_gnmt_trainer_dropout_masks(tn_gnmt_trainer *t, float **m_enc0, float **m_enc1, float **m_dec);
/* one mask of the last step (test hook): which = encoder layer 0 .. num_layers - 1 -> (B*T, dirs*H); num_layers + j -> decoder layer
 * j >= 1, (L*B, H) step-major */
int tn_gnmt_trainer_dropout_mask(tn_gnmt_trainer *t, int which, float **mask);
int tn_gnmt_trainer_adam_step(tn_gnmt_trainer *t, float lr, float beta1, float beta2, float epsilon);
int tn_gnmt_trainer_read_param(tn_gnmt_trainer *t, const char *name, int gradient, float *out_host, int64_t capacity,
                               int64_t *numel);
int tn_gnmt_trainer_destroy(tn_gnmt_trainer *t);

/* ---- device PRF1 confusion histogram -------------------------------------- */
/* Replaces the argmax + per-sample python loop of PRF1.update (reference
 * metrics/vision.py:41-49).  logits (rows,classes) fp32, labels (rows,) int32;
 * adds into mat (classes*classes) int64, row = label, col = argmax (first max). */
int tn_prf1_update(tn_ctx *ctx, const float *logits, const int32_t *labels, int rows, int classes,
                   int64_t *mat);

/* ---- multi-GPU exchange: RCCL over xGMI, one process per GPU (csrc/comm.hip) ------------
 * The path's one exchange step (BASELINE config C4).  The reference splits every DataLoader batch over its ctx list and
 * exchanges the per-frame feature rows through the file system: evaluate.py:278-281,308-321 writes
 * <root>/features/<model_id>/<video>/<frame>.npy, dataset.py:202-204 np.load()s them for the temporal / caption stage.
 * Here every rank keeps its shard of feature rows in HBM and one all-gather puts the whole (N, F) matrix on every rank.
 * Also: the sum / mean of the three trainers' flat gradient buffers (train.py:410-424 `trainer.step` over a ctx list =
 * kvstore 'device' all-reduce) and of the PRF1 confusion counts.
 *
 * Rank 0 calls tn_comm_unique_id and hands the 128 bytes to the other ranks by any out-of-band channel (a file, a TCP
 * store, MPI); every rank then calls tn_comm_create with its own context (one process per GPU, the context's device).
 * All collectives are asynchronous and ordered on the context's stream like every other entry point; librccl is opened
 * at run time (a process that already holds one - PyTorch's - shares it), a world of 1 needs none (TN_COMM_FORCE_RCCL
 * makes a single rank go through RCCL all the same: the self-test of 1-GPU boxes). */
#define TN_UNIQUE_ID_BYTES 128
#define TN_COMM_FORCE_RCCL 1
typedef struct tn_comm tn_comm;
int tn_comm_unique_id(void *id_out /* TN_UNIQUE_ID_BYTES */);
int tn_comm_create(tn_ctx *ctx, int rank, int world, const void *unique_id /* NULL for world 1 */, int flags,
                   tn_comm **out);
/* What the communicator itself reports (with RCCL behind the handle: ncclCommUserRank / ncclCommCount / ncclCommCuDevice, so a
 * record of an N-GPU run shows that RCCL saw N ranks - the reference's counterpart is len(ctx) at evaluate.py:84-85); -1 when RCCL
 * refuses the query.  tn_comm_uses_rccl: 0 for the world-1 short cut that needs no RCCL. */
int tn_comm_rank(const tn_comm *c);
int tn_comm_world(const tn_comm *c);
int tn_comm_device(const tn_comm *c);
int tn_comm_uses_rccl(const tn_comm *c);
/* out (world*rows, F) = rank-major concatenation of every rank's shard (rows, F); equal `rows` on every rank (pad the last
 * round, sharding.local_rows); shard may alias its own slot of out. */
int tn_allgather_features(tn_comm *c, const float *shard, int rows, int F, float *out);
/* in place; average != 0 divides by the world size (Trainer.step's rescale) */
int tn_allreduce_f32(tn_comm *c, float *buf, size_t n, int average);
int tn_allreduce_i64(tn_comm *c, int64_t *buf, size_t n);
int tn_comm_destroy(tn_comm *c);

/* ---- GNMT captioner: encoder + attention decoder + beam search -------------- */
/* Replaces get_gnmt_encoder_decoder / NMTModel / BeamSearchTranslator as assembled at
 * reference train_gnmt.py:223-252 and driven by evaluate() (train_gnmt.py:264-302):
 *   tn_gnmt_encode      = model.encode -> GNMTEncoder.forward (models/captioning/gnmt.py:136-160)
 *                         + decoder.init_state_from_encoder (gnmt.py:224-252)
 *   tn_gnmt_beam_search = BeamSearchTranslator.translate (utils/translation.py:55-82): the whole
 *                         decode_step / log_softmax / BeamSearchSampler loop on the device.
 * params: "<prefix>enc_rnn{i}_{l,r}_*" for the num_bi_layers bidirectional encoder layers, "<prefix>enc_rnn{i}_*" for
 * the uni-directional ones, "<prefix>dec_rnn{i}_*" ({i2h,h2h}_{weight,bias}; i < num_layers),
 * "<prefix>dec_attention_key_weight" (H,H), "<prefix>tgt_proj_{weight,bias}", "<prefix>tgt_embed_weight" (V,E).
 * cell_type gru | lstm, 2 <= num_layers <= 8, 0 <= num_bi_layers < num_layers (gnmt.py:78-80; with every layer
 * bidirectional the memory is 2H wide, which gluonnlp's Luong attention with units = H refuses), scaled_luong attention,
 * dropout off (inference); tn_gnmt_create_ex(flags = TN_GNMT_USE_RESIDUAL): use_residual (gnmt.py:155-157,394-395).
 * src (B,T,F) fp32 features, valid_len (B,) int32; samples (B,beam,max_length+2) int32 padded
 * with -1 (leading BOS), scores (B,beam) descending, valid_length (B,beam) int32;
 * *length_host = number of meaningful columns of `samples` (the width the reference returns). */
int tn_gnmt_create(tn_ctx *ctx, const tn_param *params, int n_params, const char *prefix, int cell_kind,
                   int input_size, int hidden, int embed, int vocab, int num_layers, int num_bi_layers,
                   int max_batch, int max_src_len, int beam, int max_length, tn_gnmt **out);
#define TN_GNMT_USE_RESIDUAL 1
int tn_gnmt_create_ex(tn_ctx *ctx, const tn_param *params, int n_params, const char *prefix, int cell_kind,
                      int input_size, int hidden, int embed, int vocab, int num_layers, int num_bi_layers,
                      int max_batch, int max_src_len, int beam, int max_length, int flags, tn_gnmt **out);
int tn_gnmt_encode(tn_gnmt *g, const float *src, const int32_t *valid_len, int batch, int steps, float *mem_out);
int tn_gnmt_beam_search(tn_gnmt *g, int bos, int eos, float alpha, float K, int max_length, int32_t *samples,
                        float *scores, int32_t *valid_length, int *length_host);
/* Teacher forcing: model(src, tgt[:, :-1], ...) of evaluate() (train_gnmt.py:280) ->
 * GNMTDecoder.decode_seq (gnmt.py:254-304).  tgt (B, ld) int32 DEVICE tokens, first `steps`
 * columns are fed; logits (B, steps, V) fp32.  Needs a preceding tn_gnmt_encode. */
int tn_gnmt_decode_seq(tn_gnmt *g, const int32_t *tgt, int ld, int steps, float *logits);
/* MaskedSoftmaxCELoss [EXT gluonnlp] (train_gnmt.py:256,281): loss (B,) fp32. */
int tn_masked_softmax_ce(tn_ctx *ctx, const float *logits, const int32_t *labels, int ld_labels,
                         const int32_t *valid_len, int batch, int steps, int vocab, float *loss);
int tn_gnmt_destroy(tn_gnmt *g);

#ifdef __cplusplus
}
#endif
#endif /* TENNIS_HIP_H */
