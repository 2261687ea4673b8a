/*
 * libdmv — MI355X (gfx950) kernels for the vgiwae training / evaluation hot path.
 *
 * C ABI: plain pointers and sizes, no C++ or torch types.  Every pointer named "device" is a
 * device (HBM) address owned by the caller; the library never allocates or frees device memory
 * and never synchronises the device: all work is enqueued on the `stream` argument
 * (a hipStream_t passed as void*; NULL = the default stream).
 *
 * Return value of every int function: 0 = ok, < 0 = DMV_E_* (argument / shape / workspace
 * errors), > 0 = a hipError_t from a launch.  dmv_strerror() names either.
 *
 * Each entry point cites the reference code it replaces (paths relative to the
 * vsimkus/demiss-vae checkout).
 */
#ifndef DMV_H
#define DMV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMV_VERSION 100

enum {
  DMV_OK = 0,
  DMV_E_BADARG = -1,       /* null pointer / negative size / inconsistent dims */
  DMV_E_UNSUPPORTED = -2,  /* enum value or size outside what the kernels are built for */
  DMV_E_WORKSPACE = -3,    /* caller-provided workspace too small */
};

/* activation of the residual MLP (vgiwae/shared/neural_nets.py:8-13) */
enum { DMV_ACT_RELU = 0, DMV_ACT_LEAKY_RELU = 1, DMV_ACT_TANH = 2, DMV_ACT_SIGMOID = 3 };

/* how the first layer's input row is formed from (x, m)
 * (vgiwae/models/vae.py:80-91 and the X*M products at models/iwae.py:136, models/vae.py:455) */
enum {
  DMV_IN_PLAIN = 0,         /* x                         in_dim = D   */
  DMV_IN_MASKED = 1,        /* x * m                     in_dim = D   */
  DMV_IN_CONCAT = 2,        /* [x, m]                    in_dim = 2 D */
  DMV_IN_CONCAT_MASKED = 3, /* [x * m, m]                in_dim = 2 D */
};

/* decoder likelihood (vgiwae/models/vae.py:113-129, shared/bernoulli_between_m1_and_p1.py) */
enum { DMV_GEN_NORMAL = 0, DMV_GEN_NORMAL_EPS = 1, DMV_GEN_BERN = 2, DMV_GEN_BERN_M1P1 = 3 };

/* variational family used for log q (vgiwae/shared/iwae.py:26-30, 102-132) */
enum {
  DMV_Q_OWN = 0,        /* Gaussian of the sample's own component                         */
  DMV_Q_STRATIFIED = 1, /* sum_c' pi_c' N(z; mu_c', sigma_c')  (MixtureSameFamily)        */
  DMV_Q_IMPUTATION = 2, /* (1/G) sum_g' N(z; mu_g', sigma_g')  (DeMissVAE imputation mix) */
};

/* how per-sample log-weights are combined into a bound */
enum {
  DMV_BOUND_LSE = 0,      /* logsumexp over (i, g) of log_w (+ log pi_g) - log I, mean over nz:
                             models/iwae.py:70-90 (G = 1), models/multiple_iwae.py:124-140 */
  DMV_BOUND_WEIGHTED = 1, /* sum_g pi_g (logsumexp_i log_w - log I), mean over nz:
                             models/multiple_iwae.py:142-159; with I = 1 models/multiple_vae.py:152-158 */
  DMV_BOUND_PER_COMPONENT = 2, /* mean over nz of each (b, g): models/mvbvae.py:108-109,134-137 */
};

/* Shape of a residual fully-connected network (vgiwae/shared/neural_nets.py:79-133):
 *   a0 = act(W0 x + b0); for each block: t = drop(act(W1 a + b1)); a = act(a + W2 t + b2);
 *   y = Wf a + bf.
 * Layer order everywhere below: initial, block0.linear0, block0.linear1, ..., final
 * (2*nb + 2 layers); weights are nn.Linear layout [out][in] row-major fp32. */
typedef struct DmvNet {
  int32_t in_dim;
  int32_t out_dim;
  int32_t hidden;   /* 1..512; above 256: relu only */
  int32_t nb;       /* residual blocks, 0..8 */
  int32_t act;      /* DMV_ACT_* */
  float dropout_p;  /* 0 = none */
} DmvNet;

int dmv_version(void);
const char* dmv_strerror(int rc);

/* ----- weight packing ---------------------------------------------------------------------
 * The MLP kernels read weights in MFMA-fragment order (both W and W^T, zero padded to 32-wide
 * tiles).  dmv_mlp_pack rewrites the 2*nb+2 nn.Linear tensors into `packed`
 * (dmv_mlp_packed_floats(net) floats).  Call it again whenever the weights change. */
size_t dmv_mlp_packed_floats(const DmvNet* net);
int dmv_mlp_pack(const DmvNet* net, const float* const* weights /* host array of device ptrs */,
                 const float* const* biases, float* packed /* device */, void* stream);

/* padded widths used for activations and outputs: 32*ceil(n/32) */
int32_t dmv_pad32(int32_t n);

/* ----- forward ----------------------------------------------------------------------------
 * Replaces ResidualFCNetwork.forward (vgiwae/shared/neural_nets.py:127-133) on `rows` rows.
 *   x [rows][ldx] fp32, m [rows][ldm] uint8 (only for the masked / concat input modes, D = x width)
 *   y [rows][ldy], ldy >= dmv_pad32(out_dim): columns >= out_dim are written as zeros
 *   saved: NULL for inference, or dmv_mlp_saved_floats(net, rows) floats that dmv_mlp_backward reads
 *          (opaque; sized for rows rounded up to whole 32-row wave tiles)
 *   Dropout inside the residual blocks (ResidualBlock.forward, vgiwae/shared/neural_nets.py:67-72: kept units are
 *   scaled by 1/(1-p)) takes its keep decisions from ONE of
 *     drop_keep: nb keep-masks [nb][rows][hidden] uint8 (1 = keep) - recorded masks, parity runs;
 *     drop_rng:  the library's Philox stream, drawn inside the kernels (relu networks): no mask array exists.  The
 *                caller reserves dmv_mlp_dropout_rng_elements(net, rows) counter elements starting at `offset` (a
 *                multiple of 4); *offset_base (nullable, device memory) is added at run time, which keeps a step
 *                recorded in a HIP graph random.  dmv_mlp_dropout_keep writes the same decisions out as bytes.
 *   Both NULL: no dropout (inference; with saved != NULL and dropout_p > 0 that is an error). */
typedef struct DmvDropRng {
  uint64_t seed;
  uint64_t offset;
  const uint64_t* offset_base;
} DmvDropRng;
size_t dmv_mlp_saved_floats(const DmvNet* net, int64_t rows);
size_t dmv_mlp_dropout_rng_elements(const DmvNet* net, int64_t rows);
int dmv_mlp_dropout_keep(const DmvNet* net, int64_t rows, const DmvDropRng* drop_rng, uint8_t* keep, void* stream);
int dmv_mlp_forward(const DmvNet* net, const float* packed, int32_t in_mode, int32_t D,
                    const float* x, int64_t ldx, const uint8_t* m, int64_t ldm, int64_t rows,
                    float* y, int64_t ldy, float* saved, const uint8_t* drop_keep, const DmvDropRng* drop_rng,
                    void* stream);

/* ----- backward ---------------------------------------------------------------------------
 * Replaces autograd's backward of the same network: given gy [rows][ldgy] (ldgy >= pad32(out_dim),
 * columns >= out_dim must be zero) computes
 *   dx [rows][lddx] (lddx >= pad32(in_dim); NULL = not needed), and, when grad_w != NULL, the
 *   parameter gradients, written (beta = 0) or accumulated (beta = 1) into the 2*nb+2 tensors
 *   grad_w[l] ([out][in]) and grad_b[l].
 * ws: workspace of dmv_mlp_backward_ws_bytes(net, rows) bytes.
 * drop_keep: what the forward call was given; DMV_DROP_GENERATED when it drew its decisions itself (drop_rng; relu
 * networks: the backward reads kept-and-active from the sign bits the forward saved). */
#define DMV_DROP_GENERATED ((const uint8_t*)(uintptr_t)1)
size_t dmv_mlp_backward_ws_bytes(const DmvNet* net, int64_t rows);
int dmv_mlp_backward(const DmvNet* net, const float* packed, const float* saved,
                     const uint8_t* drop_keep, const float* gy, int64_t ldgy, int64_t rows,
                     float* dx, int64_t lddx, float* const* grad_w, float* const* grad_b, float beta,
                     void* ws, size_t ws_bytes, void* stream);

/* The weight-gradient half of dmv_mlp_backward on its own, after a call of dmv_mlp_backward with grad_w == NULL and the
 * same saved / gy / rows / ws (that call leaves the per-layer output gradients in ws).  Lets a caller put the
 * weight-gradient GEMMs on a second stream while the backward pass continues upstream on the first. */
int dmv_mlp_backward_wgrad(const DmvNet* net, const float* saved, const float* gy, int64_t ldgy, int64_t rows,
                           float* const* grad_w, float* const* grad_b, float beta, void* ws, size_t ws_bytes,
                           void* stream);

/* The same for two networks in one pair of launches (the encoder and the decoder of a step whose input-gradient chains
 * are done, e.g. after dmv_vae_backward): the weight gradients that autograd forms for both networks' nn.Linear layers
 * in the reference (vgiwae/shared/neural_nets.py:101-133 under loss.backward()).  Every argument as in
 * dmv_mlp_backward_wgrad, once per network.  The two networks share one round of the GPU's wave slots, so the row
 * slices are longer than in two separate calls: the same sums in another fixed order (reproducible run to run, equal to
 * the separate calls to rounding).  Networks with more than 24 layers between them: DMV_E_UNSUPPORTED (call
 * dmv_mlp_backward_wgrad twice). */
int dmv_mlp_backward_wgrad_pair(const DmvNet* net_a, const float* saved_a, const float* gy_a, int64_t ldgy_a, int64_t rows_a,
                                float* const* grad_w_a, float* const* grad_b_a, void* ws_a, size_t ws_bytes_a,
                                const DmvNet* net_b, const float* saved_b, const float* gy_b, int64_t ldgy_b, int64_t rows_b,
                                float* const* grad_w_b, float* const* grad_b_b, void* ws_b, size_t ws_bytes_b, float beta,
                                void* stream);

/* ----- objective --------------------------------------------------------------------------
 * Sample rows are ordered n = (s*B + b)*G + g with s = z*I + i  (reference tensor shapes
 * [nz, I, B, (G,) .]).  Encoder-output addressing (in floats, per datapoint b and component g):
 *   mu(b,g,j)  = enc[b*enc_sb + g*enc_sg + j]
 *   raw(b,g,j) = enc[b*enc_sb + g*enc_sg + raw_off + j],  sigma = softplus(raw) (+1e-5 if q_eps)
 *   logit(b,g) = enc[b*enc_sb + logit_off + g]            (stratified mixtures only)            */
typedef struct DmvObjCfg {
  int32_t B, G, nz, I;
  int32_t D, Zd;
  int32_t gen_dist;        /* DMV_GEN_*  */
  int32_t q_eps;           /* 1: variational sigma = softplus(raw) + 1e-5 */
  int32_t x_per_component; /* 0: X, M are [B][D]; 1: X, M are [B][G][D] */
  int64_t enc_sb, enc_sg, raw_off, logit_off;
} DmvObjCfg;

/* z = mu + sigma*eps (vgiwae/models/vae.py:303; Normal.rsample / StratifiedMixtureSameFamily.rsample
 * at shared/stratified_mixture_same_family.py:16-22).  eps, Z, eps_out: [N][ldz], N = nz*I*B*G,
 * ldz >= Zd; columns in [Zd, ldz) are zeroed.  eps == NULL: standard normals from Philox4x32-10
 * (stream `seed`, element counter `offset` + n*ldz + j); eps_out (nullable) receives the noise used.
 * comp_idx (nullable, int64 [N], cfg->G must be 1): draw row n from component comp_idx[n] of the
 * stratified encoder output (sampling the mixture without stratification, models/vae.py:298-301).
 * offset_base (every Philox-consuming entry point has it; nullable): a device-resident counter, a multiple of 4,
 * whose value at execution time is added to `offset`.  A step recorded in a HIP graph bakes `offset` in; advancing
 * the counter on the device between replays (the host side records that increment in the graph too) gives every
 * replay fresh draws. */
int dmv_latent_sample(const DmvObjCfg* cfg, const float* enc, const float* eps, uint64_t seed, uint64_t offset,
                      const uint64_t* offset_base, int64_t ldz, const int64_t* comp_idx, float* Z, float* eps_out,
                      void* stream);

/* Encoder -> latent sample -> decoder of a small batch in ONE launch (vgiwae/models/vae.py:285-318, vae_forward:
 * predict_var_latent_params :80-91, rsample :303, generator_network :309; shared/neural_nets.py:127-133).  Equivalent to
 * dmv_mlp_forward(enc) + dmv_latent_sample + dmv_mlp_forward(dec) and writes the same buffers with the same values:
 * enc_out [enc_rows][pad32(enc out_dim)], Z / eps_out [N][ldz], dec_out [N][pad32(dec out_dim)], and - when both are
 * given - the saved activations of the two networks (dmv_mlp_saved_floats each).  enc_rows = B (cfg->x_per_component
 * == 0: one encoder row per datapoint, X / M [B][D]) or B*G (x_per_component == 1: one per imputation, X / M [B][G][D]);
 * N = nz*I*B*G <= 4096 sample rows.  cfg's encoder-output addressing must be that of a row-major
 * [enc_rows][pad32(out_dim)] array.  Each 16-row tile of samples recomputes the encoder rows it draws from, so the
 * encoder outputs and the latents never leave the compute unit between the three stages.
 * n_prior > 0 (x_per_component == 1 only; the samplers' prior-replenish proposals, vgiwae/shared/vae_imputation.py:
 * 419-441): the last n_prior of a datapoint's G components have no encoder row (X / M are [B][G - n_prior][D]); their
 * samples are drawn from mu = 0, sigma = softplus(prior_raw) (+1e-5 if q_eps).  enc_out keeps its [B][G] rows: the
 * caller fills the prior rows once, the launch writes the others.  Forward only (no saved activations).
 * dmv_vae_forward_fits: 1 when the shapes qualify (relu networks of equal hidden tile count <= 8, N <= 4096,
 * Zd <= 256), else 0; dmv_vae_forward returns DMV_E_UNSUPPORTED for shapes that do not, and for a training pass
 * (saved activations requested) of networks with dropout, whose keep decisions the separate launches draw. */
int dmv_vae_forward_fits(const DmvObjCfg* cfg, const DmvNet* enc, const DmvNet* dec);
int dmv_vae_forward(const DmvObjCfg* cfg, const DmvNet* enc, const float* enc_packed, int32_t in_mode, const float* X,
                    const uint8_t* M, const DmvNet* dec, const float* dec_packed, const float* eps, uint64_t seed,
                    uint64_t offset, const uint64_t* offset_base, int32_t n_prior, float prior_raw, float* enc_out,
                    float* enc_saved, float* Z, int64_t ldz, float* eps_out, float* dec_out, float* dec_saved,
                    void* stream);

/* Per-sample terms (vgiwae/shared/iwae.py:17-45, 66-90, 102-132; models/mvbvae.py:83-137):
 *   terms[0][n] = sum_d M_d log p(x_d|z)      terms[1][n] = sum_d log p(x_d|z)
 *   terms[2][n] = sum_j log N(z_j;0,1)        terms[3][n] = log q_own(z)   terms[4][n] = log q_mix(z)
 * dec [N][ld_dec] decoder outputs, Z [N][ldz].  q_kind DMV_Q_OWN leaves terms[4] = terms[3];
 * otherwise the mixture has mix_C components taken from the encoder output of the row's datapoint.
 * terms: 5*N floats.  comp_lp (nullable, [N][mix_C]) receives each component's log density
 * (needed by dmv_obj_backward_latent when the mixture enters the loss). */
int dmv_obj_terms(const DmvObjCfg* cfg, int32_t q_kind, int32_t mix_C, const float* dec, int64_t ld_dec,
                  const float* X, const uint8_t* M, const float* Z, int64_t ldz, const float* enc,
                  float* terms, float* comp_lp, void* stream);

/* The objective forward in ONE launch: the per-sample terms of dmv_obj_terms, the bound of dmv_obj_bound (or the
 * per-(b,g) objectives of dmv_obj_combine) and the weights the backward needs, for a block of datapoints at a time, with
 * the logsumexp over the importance / component axes done in LDS (vgiwae/shared/iwae.py:9-132, models/iwae.py:70-123,
 * models/multiple_iwae.py:124-199, models/multiple_vae.py:109-160, models/mvbvae.py:83-137).  Inputs as dmv_obj_terms;
 * every output pointer is optional except the one the chosen kind defines:
 *   kind DMV_BOUND_LSE / _WEIGHTED: bound [B] (+ mean_out scalar and neg_mean_out = -mean, the training loss of
 *        models/iwae.py:45), log_w [N], wgt [N], zscale [N], dlogit [B][G]
 *        (written in full, zeros where the bound has no pi factor), with the meaning they have in dmv_obj_bound;
 *   kind DMV_BOUND_PER_COMPONENT (I == 1): objA / objB [B][G] = mean_z sum_t coefA/B[t] * terms[t] (+ meanA/B_out scalars);
 *   kind -1: terms only.
 *   terms [5][N] and comp_lp [N][mix_C] are written when given.
 * The means (mean_out, neg_mean_out, meanA/B_out) take a second, single-workgroup launch - unless the caller brings
 * `ticket`: one uint32 in device memory that is zero when the call starts.  The workgroup that finishes last (it draws
 * the last ticket) then sums the per-datapoint results inside the same launch, in a fixed order, and leaves the counter
 * at zero again (launches of more than 256 workgroups ignore the ticket and take the second launch: the device-scope
 * release every workgroup then pays costs more than it saves).  Calls that share a ticket must not overlap in time
 * (one ticket per stream).
 * Limit: nz * I * G floats per datapoint must fit the workgroup's 60 KB of LDS (about 15 000); beyond it the call
 * returns DMV_E_UNSUPPORTED and the caller streams with dmv_obj_terms + dmv_lse_update. */
typedef struct DmvObjFwd {
  int32_t q_kind, mix_C;
  int32_t kind, use_logpi, dreg;
  float coefA[5], coefB[5];
  float *terms, *comp_lp, *log_w;
  float *bound, *mean_out, *wgt, *zscale, *dlogit;
  float *objA, *objB, *meanA_out, *meanB_out;
  float *neg_mean_out;
  uint32_t* ticket;   /* nullable */
} DmvObjFwd;
int dmv_obj_forward(const DmvObjCfg* cfg, const DmvObjFwd* f, const float* dec, int64_t ld_dec, const float* X,
                    const uint8_t* M, const float* Z, int64_t ldz, const float* enc, void* stream);

/* dmv_vae_forward and the bound of dmv_obj_forward in ONE launch (vgiwae/models/vae.py:285-318 vae_forward, then
 * vgiwae/shared/iwae.py:9-55 and vgiwae/models/iwae.py:70-123: the IWAE / ELBO-by-samples bound with its logsumexp over
 * the importance samples and mean over the outer ones; vgiwae/models/multiple_iwae.py:124-199: the tight MissSIWAE bound
 * of a stratified mixture posterior, logsumexp over (i, g) of log w + log pi_g).  Tiles are datapoint-major - the
 * nz*I*G sample rows of 16 / (nz*I*G) datapoints - so a datapoint's logsumexp stays inside a workgroup.  Every buffer of
 * dmv_vae_forward is written with the same values.  Of obj: kind = DMV_BOUND_LSE, bound [B], optional log_w / wgt /
 * zscale [N], mean_out / neg_mean_out (these need `ticket` and at most 256 tiles), and either
 *   q_kind = DMV_Q_OWN with cfg->G == 1 and use_logpi = 0, or
 *   q_kind = DMV_Q_STRATIFIED with mix_C = cfg->G > 1, use_logpi = 1, dreg = 0; then also the optional terms [5][N],
 *   comp_lp [N][mix_C] and dlogit [B][G] as dmv_obj_forward leaves them.
 * X, M are the objective's data as well as the encoder's input.  DMV_E_UNSUPPORTED for anything else (imputation
 * mixtures, the looser bound, nz*I*G > 16, objA / objB requested, means without a ticket): then call dmv_vae_forward and
 * dmv_obj_forward. */
int dmv_vae_forward_bound(const DmvObjCfg* cfg, const DmvNet* enc, const float* enc_packed, int32_t in_mode, const float* X,
                          const uint8_t* M, const DmvNet* dec, const float* dec_packed, const float* eps, uint64_t seed,
                          uint64_t offset, const uint64_t* offset_base, float* enc_out, float* enc_saved, float* Z,
                          int64_t ldz, float* eps_out, float* dec_out, float* dec_saved, const DmvObjFwd* obj, void* stream);

/* Bounds from the terms: log_w = terms[0] + terms[2] - terms[q] (q = 3 for DMV_Q_OWN else 4), combined
 * per datapoint by `kind` (DMV_BOUND_LSE / DMV_BOUND_WEIGHTED); use_logpi adds log pi_g inside the
 * logsumexp (models/multiple_iwae.py:127).  One wavefront per datapoint.
 *   bound [B] per-datapoint bounds; mean_out (nullable) their mean;
 *   log_w (nullable) [N]; wgt (nullable) [N] = d(sum_b bound_b)/d log_w_n;
 *   zscale (nullable) [N] = softmax_i(log_w) when dreg (models/iwae.py:78-81,115-117) else 1;
 *   dlogit (nullable) [B][G] = d(sum_b bound_b)/d logit through the pi factors. */
int dmv_obj_bound(const DmvObjCfg* cfg, int32_t kind, int32_t q_kind, int32_t use_logpi, int32_t dreg,
                  const float* enc, const float* terms, float* log_w, float* bound, float* mean_out,
                  float* wgt, float* zscale, float* dlogit, void* stream);

/* DeMissVAE per-(b,g) objectives (models/mvbvae.py:108-109,134-137; I must be 1):
 *   obj[b*G+g] = mean_z sum_t coef5[t] * terms[t][n];  mean_out (nullable) = mean over (b,g). */
int dmv_obj_combine(const DmvObjCfg* cfg, const float* coef5 /* host */, const float* terms, float* obj,
                    float* mean_out, void* stream);

/* kl[b*G+g] = sum_j KL(N(mu,sigma) || N(0,1))  (models/vae.py:404-406, models/mvbvae.py:119-122) */
int dmv_kl_std_normal(const DmvObjCfg* cfg, const float* enc, float* kl, void* stream);

/* Decoder-side backward: d_dec[n][c] = gscale * gb[b(n)] * wgt[n] * (base_m * M_d + base_f) * d log p(x_d|z)/d dec[n][c]
 * for every column c < ld_ddec (zeros past the parameter count).  wgt == NULL means 1; gb (nullable, [B]) is the
 * upstream gradient of each datapoint's bound; gscale (device scalar, nullable) that of a scalar loss. */
int dmv_obj_backward_dec(const DmvObjCfg* cfg, const float* dec, int64_t ld_dec, const float* X, const uint8_t* M,
                         const float* wgt, const float* gb, const float* gscale, float base_m, float base_f,
                         float* d_dec, int64_t ld_ddec, void* stream);

/* Latent-side backward: gradient of
 *   loss = sum_n wgt[n] * (base_prior*terms[2] + base_qown*terms[3] + base_qmix*terms[4])[n]
 *          + (decoder path, given as dZ) + base_w * sum_{b,g} dlogit[b][g]*logit[b][g] (linearised)
 *          + kl_coef * sum_{b,g} kl_wgt[b][g] * KL          (kl_wgt == NULL means 1)
 * with respect to the encoder output, through z = mu + sigma*eps (times zscale[n] when use_zscale)
 * and, when own_live / mix_live, through q's own parameters (no "sticking the landing").
 * gb (nullable, [B]): upstream gradient of each datapoint's bound; it multiplies wgt[n] and dlogit[b][.] of that
 * datapoint (dZ carries it already when the decoder-side call was given the same gb; the KL term has kl_wgt).
 * gscale (nullable device scalar): upstream gradient of a scalar loss; multiplies what gb multiplies and the KL term,
 * not dZ (give the decoder-side call the same gscale).
 * d_enc uses the same addressing as enc; every entry of it is written (padding entries as zeros). */
typedef struct DmvLatentBwd {
  float base_prior, base_qown, base_qmix, base_w, kl_coef;
  int32_t own_live, mix_live, use_zscale;
  int32_t mix_kind; /* DMV_Q_OWN (no mixture), DMV_Q_STRATIFIED (has logits) or DMV_Q_IMPUTATION */
} DmvLatentBwd;
int dmv_obj_backward_latent(const DmvObjCfg* cfg, const DmvLatentBwd* p, const float* enc, const float* Z,
                            const float* eps, int64_t ldz, const float* dZ, int64_t ld_dz, const float* wgt,
                            const float* gb, const float* zscale, const float* comp_lp, const float* qmix,
                            const float* dlogit, const float* kl_wgt, const float* gscale, float* d_enc, void* stream);

/* The input-gradient half of a small batch's backward pass in ONE launch (the backward of vae_forward,
 * vgiwae/models/vae.py:285-318, that autograd builds for the reference): decoder chain -> gradient w.r.t. the encoder
 * outputs -> encoder chain, equivalent to dmv_mlp_backward(dec, grad_w = NULL) + dmv_obj_backward_latent +
 * dmv_mlp_backward(enc, grad_w = NULL).  d_dec [N][ld_ddec] is the gradient w.r.t. the decoder outputs
 * (dmv_obj_backward_dec); d_enc [enc_rows][pad32(enc out_dim)] receives the gradient w.r.t. the encoder outputs (every
 * entry written); dec_ws / enc_ws (dmv_mlp_backward_ws_bytes each) receive the per-layer gradients, after which
 * dmv_mlp_backward_wgrad(dec, ..., d_dec, ..., dec_ws) and dmv_mlp_backward_wgrad(enc, ..., d_enc, ..., enc_ws) form the
 * weight gradients.  p, wgt, gb, zscale, kl_wgt, comp_lp, qmix, dlogit, gscale as in dmv_obj_backward_latent.  Without a
 * mixture term (p->base_qmix == 0): one Gaussian per encoder row (cfg->G == 1 or cfg->x_per_component == 1) and at most
 * 16 samples per encoder row (nz*I <= 16: a row's samples share a 16-row tile).  With one (stratified or imputation
 * mixture, p->mix_kind): all nz*I*G sample rows of a datapoint share a tile (nz*I*G <= 16).  In both cases the shapes
 * dmv_vae_forward takes: dmv_vae_backward_fits(cfg, enc, dec, mixture) tells; otherwise DMV_E_UNSUPPORTED.
 * dg (nullable): the arguments of dmv_obj_backward_dec; when given, the launch forms d_dec itself first (and writes it
 * out for the weight-gradient GEMMs) instead of reading it: one launch less.  Decoder output rows of up to a few
 * hundred floats (the tile's d_dec rows are staged in LDS); wider ones return DMV_E_UNSUPPORTED. */
typedef struct DmvDecGrad {
  const float* dec;        /* decoder outputs [N][ld_dec] */
  int64_t ld_dec;
  const float* X;          /* as in dmv_obj_backward_dec */
  const uint8_t* M;
  const float* wgt;        /* nullable [N] */
  const float* gb;         /* nullable [B] */
  float base_m, base_f;
} DmvDecGrad;
int dmv_vae_backward_fits(const DmvObjCfg* cfg, const DmvNet* enc, const DmvNet* dec, int32_t mixture);
int dmv_vae_backward(const DmvObjCfg* cfg, const DmvLatentBwd* p, const DmvDecGrad* dg, const DmvNet* dec,
                     const float* dec_packed, const float* dec_saved, const uint8_t* dec_drop_keep, float* d_dec, int64_t ld_ddec,
                     void* dec_ws, size_t dec_ws_bytes, const DmvNet* enc, const float* enc_packed,
                     const float* enc_saved, const uint8_t* enc_drop_keep, float* d_enc, void* enc_ws,
                     size_t enc_ws_bytes, const float* enc_out, const float* Z, const float* eps, int64_t ldz,
                     const float* wgt, const float* gb, const float* zscale, const float* kl_wgt, const float* comp_lp,
                     const float* qmix, const float* dlogit, const float* gscale, void* stream);

/* Backward from per-term cotangents, for objectives composed by the caller out of the five terms of dmv_obj_terms
 * (the importance-weighted DeMissVAE family, vgiwae/models/mvbiwae.py:97-565, combines them in a dozen ways).
 * Decoder side:  d_dec[n][c] = (g_masked[n] * M_d + g_full[n]) * d log p(x_d|z)/d dec[n][c]
 *   with g_masked / g_full [N] = d loss / d terms[0] / terms[1].
 * Latent side: gradient w.r.t. the encoder output of
 *   sum_n g_prior[n]*terms[2][n] + g_qown[n]*terms[3][n] + g_qmix[n]*terms[4][n]  + (decoder path, given as dZ)
 *   through z = mu + sigma*eps (all of it times zscale[n] when zscale is given: the weight hook of the doubly
 *   reparameterised estimator, models/mvbiwae.py:103-114) and, when q_live, through q's own parameters.
 *   mix_kind DMV_Q_IMPUTATION or DMV_Q_STRATIFIED; comp_lp / qmix as produced by dmv_obj_terms. */
int dmv_obj_backward_dec_terms(const DmvObjCfg* cfg, const float* dec, int64_t ld_dec, const float* X, const uint8_t* M,
                               const float* g_masked, const float* g_full, float* d_dec, int64_t ld_ddec, void* stream);
int dmv_obj_backward_latent_terms(const DmvObjCfg* cfg, int32_t mix_kind, int32_t q_live, const float* enc,
                                  const float* Z, const float* eps, int64_t ldz, const float* dZ, int64_t ld_dz,
                                  const float* g_prior, const float* g_qown, const float* g_qmix, const float* zscale,
                                  const float* comp_lp, const float* qmix, float* d_enc, void* stream);

/* Latent-side backward for a mixture encoder sampled ancestrally with implicit reparameterisation gradients
 * (vgiwae/shared/reparametrized_mixture_same_family.py:27-191 rsample + _distributional_transform; used by the
 * reparametrised_mixture* variational families, vgiwae/models/vae.py:108-260).  Rows n = s*B + b (G must be 1),
 * encoder output addressed like a stratified mixture of mix_C components.  Computes the gradient, w.r.t. the
 * encoder output (all components' mu, raw sigma and the logits), of
 *   sum_n <g_n, z_n>,  g_n = dZ[n] + wgt[n] * (base_prior * d log p(z)/dz + base_qmix * d log q_mix(z)/dz)
 * through z only with mix_live 0 (the "sticking the landing" form: q's own parameters are not differentiated
 * directly); with mix_live 1 the term sum_n wgt[n] * base_qmix * d log q_mix(z_n; theta)/d theta at fixed z_n is
 * added (vgiwae/models/iwae.py:104-106: the score is taken on the live distribution unless STL or DREG is set).
 * ws: dmv_obj_implicit_ws_floats(cfg, mix_C) floats.  d_enc: written for every addressed entry. */
size_t dmv_obj_implicit_ws_floats(const DmvObjCfg* cfg, int32_t mix_C);
int dmv_obj_backward_latent_implicit(const DmvObjCfg* cfg, int32_t mix_C, const float* enc, const float* Z,
                                     int64_t ldz, const float* dZ, int64_t ld_dz, const float* wgt,
                                     float base_prior, float base_qmix, int32_t mix_live, float* ws, float* d_enc,
                                     void* stream);

/* Marginal log-probability of incomplete (and complete) datapoints by numerical integration over a latent grid
 * (vgiwae/shared/vae_marginal_logprob.py:194-307, get_marginal_logprob_for_Zgrid): dec [Nz][ld_dec] are the decoder
 * outputs at the Nz grid points, logw [Nz] = log p(z) + log(cell volume).
 *   out_obs[b]  = logsumexp_z(logw_z + sum_d M_bd log p(x_bd | z));   out_full[b] (nullable): the same without M.
 * ws: dmv_grid_marginal_ws_floats(Nz, D) floats. */
size_t dmv_grid_marginal_ws_floats(int64_t Nz, int32_t D);
int dmv_grid_marginal(int32_t gen_dist, int32_t D, int64_t Nz, const float* dec, int64_t ld_dec, const float* logw,
                      int32_t B, const float* X, const uint8_t* M, float* ws, float* out_obs, float* out_full,
                      void* stream);

/* Streaming logsumexp for the K-sample log-likelihood (test/refit_encoder_using_iwae_and_estimate_loglik.py
 * :189-222): folds a chunk of Kc samples per datapoint (terms of that chunk, rows k*B + b) into
 * run_max / run_sum [B] (initialise to -inf / 0).  log-lik_b = run_max + log(run_sum) - log(K). */
int dmv_lse_update(int32_t B, int64_t Kc, const float* terms, int32_t q_term /* 3 own, 4 mixture */,
                   float* run_max, float* run_sum, void* stream);

/* ----- imputation update -------------------------------------------------------------------
 * Persistent imputation buffer rows (vgiwae/utils/imputation_distribution.py:145-157 gather,
 * :180-187 scatter).  idx: int64 row numbers; rows are D floats.  Bit-exact copies. */
int dmv_rows_gather(const float* src, const int64_t* idx, int64_t n, int32_t D, float* dst, void* stream);
/* Rows sel[i] of up to four arrays at once (the arrays of a data split - data, mask, targets, indices - each with its
 * own row length in bytes): one launch builds the whole batch tuple that the reference's DataLoader collates on the
 * host (vgiwae/data/missing_data_module.py:289-312).  dst[k] receives n_rows rows of row_bytes[k] bytes. */
int dmv_rows_gather_multi(int32_t n_arrays, const void* const* src, void* const* dst, const int64_t* row_bytes,
                          const int64_t* sel, int64_t n_rows, void* stream);
int dmv_rows_scatter(float* dst, const int64_t* idx, int64_t n, int32_t D, const float* src, void* stream);

/* Sampling tail of one imputation iteration for N = B*K rows (vgiwae/shared/vae_imputation.py:38-52
 * pseudo-Gibbs; :526-561 and :657-666 LAIR):
 *   x_m ~ p(. | dec[row_map ? row_map[n] : n])   Gaussian: m + s*noise; Bernoulli: 1[noise < sigmoid(logit)]
 *   clamp to [clip_min, clip_max] (NaN-propagating, like torch.max/min), non-finite -> X_old,
 *   X_out = M ? X_old : x_m            (clip_after_blend = 1: the LAIR order, clamp applied to the blend)
 *   optional write-back: buf[idx_imp[n]] = X_out[n].
 * noise == NULL: Philox stream (seed, offset + n*D + d). */
int dmv_impute_tail(int64_t N, int32_t D, int32_t gen_dist, const float* dec, int64_t ld_dec,
                    const int64_t* row_map, const float* X_old, const uint8_t* M, const float* noise,
                    uint64_t seed, uint64_t offset, const uint64_t* offset_base, const float* clip_min,
                    const float* clip_max, int32_t clip_after_blend, float* X_out, float* buf, const int64_t* idx_imp,
                    void* stream);

/* K categorical draws per datapoint from softmax(logits[b][0..P)) by inverse CDF
 * (replaces Categorical(logits).sample((K,)), vgiwae/shared/vae_imputation.py:185-186, :679-680).
 * u (nullable) [B][K] uniforms in [0,1); NULL: Philox stream (seed, offset + b*K + k).  idx int64 [B][K]. */
int dmv_categorical_draw(int32_t B, int32_t P, int32_t K, const float* logits, const float* u, uint64_t seed,
                         uint64_t offset, const uint64_t* offset_base, int64_t* idx, void* stream);

/* The same draws when the logits are the log-weights of proposals held as a [groups][B][inner] array (LAIR: groups =
 * importance samples i, or (iteration, i) pairs for the final resampling, vgiwae/shared/vae_imputation.py:181-198,
 * 671-694; inner = K + r components), without first permuting them to [B][groups * inner]: category p = group * inner + k
 * of datapoint b reads logits[(group * B + b) * inner + k].  idx (nullable) int64 [B][K] receives p; rows (nullable)
 * int64 [B][rows_keep] receives, for the first rows_keep draws of each datapoint, (group * B + b) * inner + k itself:
 * the row of the picked proposal in every array ordered like the logits (decoder outputs, latents), which is what the
 * gathers that follow need (dmv_impute_tail's row_map, dmv_rows_gather). */
int dmv_categorical_draw_grouped(int32_t B, int32_t groups, int32_t inner, int32_t K, const float* logits, const float* u,
                                 uint64_t seed, uint64_t offset, const uint64_t* offset_base, int64_t* idx, int64_t* rows,
                                 int32_t rows_keep, void* stream);

/* LAIR's whole imputation update in one launch (vgiwae/shared/vae_imputation.py:390-563 per iteration, :671-694 the final
 * resampling; DMIS log-weights shared/iwae.py:57-100; global multinomial resampling :181-198), one importance sample per
 * component.  X [B*K][D] holds the K imputations of every datapoint and is updated in place, M [B*K][D] their masks.
 * Per iteration t < T: encoder on the imputations (in_mode as dmv_mlp_forward), proposals from the K components and
 * n_prior standard-normal ones (raw sigma prior_raw), decoder, log-weights against the uniform mixture of the K + n_prior
 * components, K + n_prior categorical draws per datapoint of which the first K are kept, new imputations from the chosen
 * proposals' decoder outputs (clamped after blending when clip_min / clip_max are given).  final_resample: afterwards K
 * proposals per datapoint are drawn from the pooled T (K + n_prior) of all iterations, decoded once more and sampled
 * without clamping.  buf / idx_imp (nullable): the imputation buffer rows that receive the final imputations.
 * Noise: eps [T][B*(K+n_prior)][Zd], draws [T][B][K+n_prior] (proposal numbers), final_draws [B][K] (pooled numbers
 * t*(K+n_prior) + c), noise [T (+1)][B*K][D] - each nullable; what is null comes from the Philox stream `seed` at the
 * counters off_eps[t] + (b*(K+n_prior)+k)*Zd + j, off_draw[t] + b*(K+n_prior) + k (off_draw[T] + b*K + k for the final
 * draws), off_tail[t] + (b*K+k)*D + d (host arrays of T, T+1, T+1 entries; offset_base as elsewhere): the ranges the
 * separate launches (dmv_vae_forward, dmv_obj_forward, dmv_lair_resample_tail) would consume, so both ways use the same
 * stream.  Everything between the imputations stays in LDS: relu networks of equal hidden width <= 256, K + n_prior <= 16,
 * T <= 32 (dmv_lair_update_fits; DMV_E_UNSUPPORTED otherwise: use the separate launches).  No dropout is applied.
 * lw_out (nullable) [T][B][K+n_prior] receives every iteration's log-weights. */
int dmv_lair_update_fits(const DmvNet* enc, const DmvNet* dec, int32_t B, int32_t K, int32_t n_prior, int32_t T, int32_t D,
                         int32_t Zd, int32_t final_resample);
int dmv_lair_update(const DmvNet* enc, const float* enc_packed, int32_t in_mode, const DmvNet* dec, const float* dec_packed,
                    int32_t B, int32_t K, int32_t n_prior, int32_t T, int32_t D, int32_t Zd, int32_t gen_dist, int32_t q_eps,
                    float prior_raw, int32_t final_resample, float* X, const uint8_t* M, const float* eps,
                    const int64_t* draws, const int64_t* final_draws, const float* noise, uint64_t seed,
                    const uint64_t* off_eps, const uint64_t* off_draw, const uint64_t* off_tail, const uint64_t* offset_base,
                    const float* clip_min, const float* clip_max, float* buf, const int64_t* idx_imp, float* lw_out,
                    void* stream);

/* All iterations of the pseudo-Gibbs sampler in one launch (vgiwae/shared/vae_imputation.py:26-82: encoder on the
 * imputation, z ~ q(. | x), decoder, x_m ~ p(. | z) clamped, blended with the observed values - T times): the same kernel
 * with every imputation row a datapoint of its own.  X [rows][D] is updated in place; eps [T][rows][Zd], noise
 * [T][rows][D] nullable (then Philox at off_eps[t] + r*Zd + j and off_tail[t] + r*D + d, the ranges dmv_vae_forward and
 * dmv_impute_tail would take); buf / idx_imp as above.  Same values as those launches; same limits as dmv_lair_update. */
int dmv_pgibbs_update_fits(const DmvNet* enc, const DmvNet* dec, int32_t rows, int32_t T, int32_t D, int32_t Zd);
int dmv_pgibbs_update(const DmvNet* enc, const float* enc_packed, int32_t in_mode, const DmvNet* dec, const float* dec_packed,
                      int32_t rows, int32_t T, int32_t D, int32_t Zd, int32_t gen_dist, int32_t q_eps, float* X,
                      const uint8_t* M, const float* eps, const float* noise, uint64_t seed, const uint64_t* off_eps,
                      const uint64_t* off_tail, const uint64_t* offset_base, const float* clip_min, const float* clip_max,
                      float* buf, const int64_t* idx_imp, void* stream);

/* LAIR's resampling and sampling tail in one launch (vgiwae/shared/vae_imputation.py:181-198, 521-561): per datapoint
 * K_draws draws as dmv_categorical_draw_grouped (uniforms u or the Philox stream (draw_seed, draw_offset + b*K_draws + k)),
 * the first K kept, then the K new imputation rows as dmv_impute_tail with clip_after_blend = 1 computes them from the
 * picked proposals' decoder outputs (noise or (tail_seed, tail_offset + n*D + d); X_old, M, X_out [B*K][D]; optional
 * write-back into buf).  rows (nullable) int64 [B][K] receives the picked rows.  Same values as the two launches.
 * At most 1024 proposals per datapoint and 64 draws (DMV_E_UNSUPPORTED beyond: use the two launches). */
int dmv_lair_resample_tail(int32_t B, int32_t groups, int32_t inner, int32_t K_draws, int32_t K, const float* logits,
                           const float* u, uint64_t draw_seed, uint64_t draw_offset, int32_t D, int32_t gen_dist,
                           const float* dec, int64_t ld_dec, const float* X_old, const uint8_t* M, const float* noise,
                           uint64_t tail_seed, uint64_t tail_offset, const uint64_t* offset_base, const float* clip_min,
                           const float* clip_max, float* X_out, float* buf, const int64_t* idx_imp, int64_t* rows,
                           void* stream);

/* Rows of the proposal table selected by the resampling draws (the gather of gr_resampling,
 * vgiwae/shared/vae_imputation.py:193-198, after keeping the first K draws, :521-524): draws int64 [B][Kp] hold proposal
 * numbers v = i*Kp + k; row_map int64 [B*K] receives (i*B + b)*Kp + k, the row of that proposal in tensors ordered
 * (i, b, k) such as the decoder outputs handed to dmv_impute_tail. */
int dmv_lair_row_map(int32_t B, int32_t K, int32_t Kp, const int64_t* draws, int64_t* row_map, void* stream);

/* keep[i] = 1 with probability 1-p (dropout masks of the residual blocks, neural_nets.py:65,70) */
int dmv_dropout_masks(uint8_t* keep, int64_t n, float p, uint64_t seed, uint64_t offset, const uint64_t* offset_base,
                      void* stream);

/* Divergences between the model's posterior on the latent grid and a variational distribution, per row
 * (vgiwae/shared/vae_marginal_logprob.py:91-192, the posterior-divergence metrics of the toy study):
 *   p(z) = exp(log_prior[z] + sum_d [M_d] log p(x_d|z) - log_marg[r]),   q(z) = (1/Kq) sum_k N(z; mu_rk, diag sigma_rk^2)
 *   out[0][r] = KL(p||q), out[1][r] = KL(q||p), out[2][r] = JSD(p, q), sums over the grid times the cell volume dz.
 * ws: the per-grid-point table left by dmv_grid_marginal for the same decoder outputs; Zgrid [Nz][Zd];
 * X [R][D]; M nullable (all entries observed); log_marg [R] the matching output of dmv_grid_marginal;
 * q_mu, q_sigma [R][Kq][Zd]. */
int dmv_grid_divergences(int32_t gen_dist, int32_t D, int32_t Zd, int64_t Nz, const float* ws, const float* log_prior,
                         const float* Zgrid, float dz, int32_t R, const float* X, const uint8_t* M,
                         const float* log_marg, int32_t Kq, const float* q_mu, const float* q_sigma, float* out,
                         void* stream);

/* Optimiser step (vgiwae/models/vae.py:349-390: Adam / AMSGrad with one parameter group per network, preceded by
 * clip_grad_norm_ over all parameters; shared/mvb2_manual_optimisation.py:9-11).  One call =
 *   norm = || all grads ||_2;  coef = max_norm > 0 ? min(1, max_norm / (norm + 1e-6)) : 1;  g' = coef * g   (grads are
 *   left untouched in memory);  step[group] += 1;  Adam update of every tensor with its group's hyper-parameters:
 *   m += (1-b1)(g'-m);  v = b2 v + (1-b2) g'^2;  [vmax = max(vmax, v)];
 *   p -= lr / (1-b1^t) * m / (sqrt(vmax or v) / sqrt(1-b2^t) + eps).
 * params / grads / exp_avg / exp_avg_sq / max_exp_avg_sq: host arrays of n device pointers (max_exp_avg_sq == NULL or
 * an entry NULL: no AMSGrad for that tensor); numel[i] elements each; group_of[i] in [0, n_groups).
 * groups[g].lr and .step are device scalars (float): schedulers write lr, the call increments step.
 * ws: dmv_opt_ws_floats(n, numel) floats.  norm_out (nullable device scalar) receives the gradient norm.
 * The pointers are passed to the kernels by value: a step recorded in a HIP graph replays without host copies.
 * pack_map: NULL, or n entries in device memory (dmv_mlp_pack_map fills them on the host; copy them over once): where
 * tensor i lives in its network's fragment image.  The step then also writes every new parameter value into the
 * image (entries with image == NULL are skipped), so a forward pass after the step needs no dmv_mlp_pack.
 * counter, counter_add: NULL / 0, or a device counter to which the step adds counter_add on the side (the Philox
 * offset_base of a recorded step: advancing it here saves the step a launch). */
typedef struct DmvPackMap {
  float* image;              /* the network's fragment image (dmv_mlp_pack's `packed`); NULL: not part of one */
  int64_t fwd_off, tr_off;   /* floats: forward and transposed fragment sets of a weight; bias fragments in fwd_off */
  int32_t N, K;              /* weight [N][K]; bias: N entries */
  int32_t S, Sn;             /* fragment groups per 32-wide tile of the forward / transposed set */
  int32_t is_bias, reserved;
} DmvPackMap;
/* entries for the 2*nb+2 weight and bias tensors of `net` whose image lives at `packed` */
int dmv_mlp_pack_map(const DmvNet* net, float* packed, DmvPackMap* weights, DmvPackMap* biases);
#define DMV_OPT_MAX_TENSORS 72
#define DMV_OPT_MAX_GROUPS 4
typedef struct DmvAdamGroup {
  const float* lr;
  float* step;
  double beta1, beta2, eps; /* doubles: 1 - beta is formed in double precision before rounding, as torch does */
} DmvAdamGroup;
size_t dmv_opt_ws_floats(int32_t n, const int64_t* numel);
int dmv_opt_adam_step(int32_t n, float* const* params, const float* const* grads, float* const* exp_avg,
                      float* const* exp_avg_sq, float* const* max_exp_avg_sq, const int64_t* numel,
                      const int32_t* group_of, int32_t n_groups, const DmvAdamGroup* groups, float max_norm, float* ws,
                      size_t ws_floats, float* norm_out, const DmvPackMap* pack_map, uint64_t* counter, uint64_t counter_add,
                      void* stream);

/* dmv_opt_adam_step for callers that cannot allocate device memory or bring a workspace (code running under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments
 * minus ws / ws_floats, same kernels, same results bit for bit; the workspace is a file-scope device array of
 * DMV_OPT_STATIC_WS_FLOATS floats (256 KiB, held by every GPU the library runs on until the process ends).
 *   - limit: dmv_opt_ws_floats(n, numel) <= DMV_OPT_STATIC_WS_FLOATS, i.e. about 16 300 chunks of 4096 parameters
 *     (~66 M parameters); beyond it the call returns DMV_E_WORKSPACE and touches nothing;
 *   - the call allocates and frees nothing; its first call on a device looks the array's address up
 *     (hipGetSymbolAddress), so make that first call outside any capture;
 *   - every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two host threads)
 *     race on it.  Calls ordered on one stream are safe; nothing in the array needs clearing between calls. */
#define DMV_OPT_STATIC_WS_FLOATS 65536
int dmv_opt_adam_step_static(int32_t n, float* const* params, const float* const* grads, float* const* exp_avg,
                             float* const* exp_avg_sq, float* const* max_exp_avg_sq, const int64_t* numel,
                             const int32_t* group_of, int32_t n_groups, const DmvAdamGroup* groups, float max_norm,
                             float* norm_out, const DmvPackMap* pack_map, uint64_t* counter, uint64_t counter_add,
                             void* stream);

/* Measurement aid (no counterpart in the reference): the shader clock the chip holds under a load.  dmv_clock_watch
 * enqueues one wave that stamps the shader-cycle counter and the constant-rate counter, naps until the latter has
 * advanced by `ticks` (1 .. 2^28; the wave also leaves after a fixed number of naps), and stamps again:
 * out4 = {cycles0, ticks0, cycles1, ticks1} (device memory).  Launch it on a side stream while the kernels of interest
 * run on another; clock = (cycles1 - cycles0) / (ticks1 - ticks0) x dmv_wall_clock_khz(). */
int dmv_clock_watch(int64_t ticks, uint64_t* out4, void* stream);
int dmv_wall_clock_khz(int32_t* khz);

#ifdef __cplusplus
}
#endif
#endif /* DMV_H */
