/*
 * node_hip.h -- C ABI of libnode_hip.so, the MI355X (gfx950) Neural-ODE
 * forward/adjoint integrator.
 *
 * This is the drop-in boundary for ONE path of fabiocarrara/neural-ode-features:
 *
 *     out = self.odeint(self.odefunc, x, self.integration_time,
 *                       method=self.method, rtol=self.tol, atol=self.tol)
 *                                                  -- reference model.py:367
 *
 * i.e. `torchdiffeq.odeint` / `odeint_adjoint` (imported at model.py:3, an
 * un-vendored third-party package) driving the Conv-GroupNorm-ReLU dynamics
 * `ODEfunc.forward` (model.py:339-348) built from `ConcatConv2d`
 * (model.py:313-323) and `nn.GroupNorm(min(32, C), C)` (model.py:268-271).
 *
 * Plain C: pointers and sizes only, no torch / C++ types.  The reference is
 * Python, so its FFI for this path is `ctypes`; the binding a maintainer adds
 * is shown in INTEGRATION.md and shipped as neural-ode-features_amd/_lib.py.
 *
 * Contract
 *  - every device buffer (inputs, outputs, workspace) is allocated and owned by
 *    the caller; the library never allocates or frees device memory and keeps
 *    no pointer after the call's work on `stream` has completed;
 *  - tensors are fp32, contiguous, NCHW exactly as PyTorch hands them over
 *    (the library converts to its internal NHWC layout inside the workspace);
 *  - `params` points INTO the live nn.Parameter storages (no copies), in the
 *    reference's layouts: conv weights are [C, C+1, 3, 3] with input channel 0
 *    being the time channel (model.py:321-322);
 *  - all work is enqueued on the caller's HIP stream (`hipStream_t` passed as
 *    void*).  The adaptive step loop runs WITHOUT host decisions: accept /
 *    reject, the next step size, which output times a step passed, dense
 *    output and the end of the interval are all decided by kernels; the host
 *    enqueues as many steps as the previous solve of the same problem needed,
 *    then synchronises once to read the controller record (and tops up two
 *    steps at a time if the interval is not finished; steps enqueued past the
 *    end return at once on the device).  `stats` is valid on return;
 *  - return 0 on success, a negative NODE_ERR_* otherwise; the message is
 *    available from node_last_error() (thread-local).  No exceptions, no abort.
 */
#ifndef NODE_HIP_H
#define NODE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NODE_ABI_VERSION 6

/* method -- the two solver names that reach model.py:367 on the graded configs
 * (`'dopri5'` train.py:219 default; `'rk4'` BASELINE.json configs[0]). */
enum { NODE_METHOD_DOPRI5 = 0, NODE_METHOD_RK4 = 1 };

enum {
  NODE_OK = 0,
  NODE_ERR_NULL = -1,         /* a required pointer is NULL                      */
  NODE_ERR_SHAPE = -2,        /* n/c/h/w/groups inconsistent                      */
  NODE_ERR_UNSUPPORTED = -3,  /* shape outside what the gfx950 kernels tile       */
  NODE_ERR_WORKSPACE = -4,    /* ws_bytes < node_workspace_bytes(...)             */
  NODE_ERR_MAX_STEPS = -5,    /* max_num_steps exceeded                           */
  NODE_ERR_NONFINITE = -6,    /* non-finite state / error norm                    */
  NODE_ERR_DT_UNDERFLOW = -7, /* t + dt == t  (upstream: 'underflow in dt')       */
  NODE_ERR_HIP = -8,          /* a HIP runtime call failed                        */
  NODE_ERR_ARG = -9           /* bad scalar argument (method, n_t, time order...) */
};
#define NODE_PENDING 1        /* node_stats.status of a solve enqueued with node_solve_opts.blind_steps */

/* State shape [n, c, h, w]; groups = min(32, c) (model.py:271); eps = 1e-5. */
typedef struct node_shape {
  int32_t n, c, h, w;
  int32_t groups;
  float eps;
} node_shape;

/* The ten parameters of ODEfunc in `ODEfunc.parameters()` order
 * (model.py:331-336).  Device pointers. */
typedef struct node_params {
  const float* norm1_w; /* [C] */
  const float* norm1_b; /* [C] */
  const float* conv1_w; /* [C, C+1, 3, 3], input channel 0 = time */
  const float* conv1_b; /* [C] */
  const float* norm2_w;
  const float* norm2_b;
  const float* conv2_w;
  const float* conv2_b;
  const float* norm3_w;
  const float* norm3_b;
} node_params;

/* Solver telemetry.  `nfe` follows the reference's counter (model.py:340): one
 * per ODEfunc.forward, including the recompute inside every adjoint eval. */
typedef struct node_stats {
  int32_t nfe;
  int32_t accepted;
  int32_t rejected;
  int32_t status;   /* NODE_OK or the NODE_ERR_* that stopped the solve */
  double last_dt;   /* step size proposed for the step after the last one */
  double t_final;   /* solver time reached */
  double first_dt;  /* initial step chosen (dopri5) */
} node_stats;

/* What a solve that was enqueued WITHOUT a final synchronisation (node_solve_opts.blind_steps) leaves in device
 * memory, written on the stream behind its last step.  `steps` = steps actually tried (steps enqueued past the end
 * of the interval return at once).  `miss` != 0: the steps enqueued did not finish the interval, or it stopped with
 * `status` -- its outputs must not be used. */
typedef struct node_step_record {
  int32_t done, status, steps, accepted, rejected, miss;
  double t, dt, first_dt;
  double t_prev, dt_used;   /* start and size of the last step tried (solver time: -t for a solve towards smaller t): when it was
                             * accepted, t_prev + dt_used - (end of the interval) is how far the solve overshot its end -- a small
                             * fraction of dt_used means the NEXT solve of this kind may need one step more */
} node_step_record;

/* Optional knobs (pass NULL for upstream behaviour). */
typedef struct node_solve_opts {
  int32_t max_num_steps;    /* <=0: 2^31-1 like upstream                          */
  int32_t n_forced_dt;      /* >0: replay mode -- take exactly these step sizes,  */
  const double* forced_dt;  /*     every step accepted (host pointer)             */
  int32_t record_dt;        /* >0: capacity of dt_log                             */
  double* dt_log;           /* host: dt tried at each step (<0 => rejected)       */
  int32_t* n_dt_log;        /* host: number of entries written                    */
  /* Deferred completion (dopri5, n_t == 2, no replay / dt log): enqueue exactly `blind_steps` steps and RETURN
   * WITHOUT SYNCHRONISING.  The caller keeps its queue fed across the solve and learns the outcome later from
   * `record` (device memory, filled on the stream); `miss_flag` (device float, nullable) is incremented when the
   * record says miss, so that work which commits results (node_sgd_step's skip flag) can be predicated on the
   * device.  stats then hold the counts as if all `blind_steps` had been needed (nfe = 2 + 6 blind_steps ...),
   * status NODE_PENDING; the true step count is in `record` (steps past the end of the interval return at once). */
  int32_t blind_steps;
  node_step_record* record; /* device */
  float* miss_flag;         /* device, nullable */
  /* node_solve_adjoint only: the caller keeps just the last state of the trajectory (`out[-1]`, what ODEBlock returns
   * with return_last_only, model.py:368-369), so dL/dy_out is zero in every other slice: `grad_out` then points to
   * that ONE slice [n, c, h, w] and the zero slices are neither materialised by the caller nor read here. */
  int32_t grad_last_only;
  /* GLOBAL-NORM mode of a data-parallel solve (ABI 6; dopri5; SURVEY.md 8e, collective 2): every rank integrates its shard of the
   * batch, and the sums the step controller decides from -- sum (err / tol)^2 of every state segment, the sums of Hairer's
   * initial step -- are added over the ranks before each decision, so that ALL RANKS TAKE IDENTICAL STEPS (no straggler, and for
   * the sharded segments y and adj_y exactly the mixed norm of the unsharded batch).  The library packs its local sums into
   * `norm_buf` (device, >= 8 floats) on the stream and calls `norm_reduce(ctx, norm_buf, 8, stream)`, which must leave the SUM
   * over the `norm_world` ranks in the buffer, enqueued on (or ordered behind) that stream; the next launch reads it.  Called from
   * inside node_solve_fwd / node_solve_adjoint on the calling thread: twice per interval for the initial step, once per step.
   * The segments every rank holds as its own partial (adj_params, adj_t) enter with the mean of the ranks' ratios; the fp16-pair
   * "repeat this step" flag of any rank repeats the step on all.  NULL: local norms (upstream behaviour per process). */
  void (*norm_reduce)(void* ctx, float* norm_buf, int32_t n, void* stream);
  void* norm_reduce_ctx;
  float* norm_buf;
  int32_t norm_world;
} node_solve_opts;

/* Per-kernel-class timing collected with HIP events on the caller's stream
 * (bench.py's roofline block).  Classes: 0 conv3x3 fwd/dgrad as one fused kernel,
 * 1 wgrad GEMM, 2 the component GEMMs of a conv3x3 fwd/dgrad that runs as the
 * F(4x4,3x3) pipeline (`flops` counts the convolution's, as for class 0);
 * 3..8 the HBM-bound GroupNorm / transform passes of that pipeline, one class per
 * kernel instance (3 combine, 4 forward pass, 5 forward pass + next combine,
 * 6 forward pass + backward top, 7 backward pass, 8 backward pass + next combine):
 * for these `flops` holds the ALGORITHMIC BYTES of the launches (every tensor the
 * pass must read or write, once). */
#define NODE_PROFILE_CLASSES 9
typedef struct node_profile {
  int64_t launches[NODE_PROFILE_CLASSES];
  double total_ms[NODE_PROFILE_CLASSES];
  double flops[NODE_PROFILE_CLASSES]; /* algorithmic FLOPs (classes 0..2) / bytes (3..8) of those launches */
} node_profile;

int node_abi_version(void);
const char* node_last_error(void);

/* Number of fp32 elements of the flat parameter vector (18*C*C + 26*C). */
size_t node_param_count(const node_shape* shape);

/* Bytes of caller-provided device workspace needed by the calls below.
 * adjoint != 0 sizes for node_solve_adjoint / node_odefunc_vjp. */
size_t node_workspace_bytes(const node_shape* shape, int method, int adjoint, int n_t);

/* 1 when a dopri5 forward solve of this shape runs as ONE resident launch on the current device (the bs = 1 census of
 * evaluate.py:97-142: f0, the initial step, every step with its decision and dense output inside one kernel), else 0. */
int node_solve_is_resident(const node_shape* shape);

/* f = ODEfunc(t, y)                                     -- model.py:339-348 */
int node_odefunc_fwd(const node_shape* shape, const node_params* params, float t,
                     const float* y, float* f,
                     void* ws, size_t ws_bytes, void* stream);

/* What the upstream adjoint asks autograd for at every stage:
 *   f = ODEfunc(t, y);  (vjp_t, vjp_y, vjp_params) = grad(f, (t, y, *params), cot)
 * vjp_t: 1 float (device); vjp_params: node_param_count floats (device), flat in
 * parameters() order, each parameter in its PyTorch layout. */
int node_odefunc_vjp(const node_shape* shape, const node_params* params, float t,
                     const float* y, const float* cot,
                     float* f, float* vjp_y, float* vjp_t, float* vjp_params,
                     void* ws, size_t ws_bytes, void* stream);

/* Diagnostics (tests, tools): y = conv3x3(x, weight[:, 1:], padding = 1) for an [n, c, 8, 8] (n % 8 == 0) or [n, c, 16, 16]
 * (n % 2 == 0) tensor, c % 64 == 0, through the
 * Winograd F(4x4,3x3) pipeline the solver uses when its tolerance allows (csrc/wino4.h); dgrad != 0: the data
 * gradient (transposed convolution) instead.  weight is a ConcatConv2d filter [c][c+1][3][3] (model.py:320-323). */
size_t node_conv3x3_w4_workspace_bytes(const node_shape* shape);
int node_conv3x3_w4(const node_shape* shape, const float* weight, int dgrad, const float* x, float* y,
                    void* ws, size_t ws_bytes, void* stream);

/* Diagnostics (tests): the exact three-way bf16 split (x = h + m + l) the component GEMMs of that pipeline apply to
 * their fp32 operands on the device, element by element: out[3 i + p] = part p of x[i], widened to float.
 * x: n floats, out: 3 n floats (device); n a positive multiple of 8. */
int node_w4_split3(const float* x, float* out, size_t n, void* stream);

/* Diagnostics (tests; ABI 6): what the last node_solve_adjoint of the process did with the fp16-PAIR operand format of the
 * F(4x4,3x3) pipeline (csrc/wino4.h): out[0] = 1 when the solve used it, out[1] = steps the device controller REPEATED because a
 * cotangent left the range of its power-of-two scale (such a step is not a solver step: it appears in no statistic), out[2] = the
 * cotangent exponent at the end, out[3] = 0.  Filled only by solves that ran with NODE_TUNE_W4_STATS=1 in the environment (two
 * words more in the final read-back); -1 in out[1] otherwise.  NODE_TUNE_W4_GSKEW=k (diagnostics) starts every interval with the
 * cotangent exponent k too high, so that its first step overflows and is repeated. */
int node_w4_pair_stats(int32_t* out4);

/* torchdiffeq.odeint(ODEfunc, y0, t, rtol, atol, method)  -- model.py:367
 * t_pts: host array of n_t strictly monotonic times; y_out: [n_t, n, c, h, w]
 * with y_out[0] = y0. */
int node_solve_fwd(const node_shape* shape, const node_params* params,
                   const float* y0, const float* t_pts, int n_t,
                   float rtol, float atol, int method, const node_solve_opts* opts,
                   float* y_out, node_stats* stats,
                   void* ws, size_t ws_bytes, void* stream);

/* Backward of torchdiffeq.odeint_adjoint (triggered by train.py:51):
 * y_traj = the forward's y_out, grad_out = dL/dy_out, both [n_t, n, c, h, w].
 * Outputs: grad_y0 [n, c, h, w]; grad_params [node_param_count] flat in
 * parameters() order; grad_t [n_t] or NULL (the reference discards it). */
int node_solve_adjoint(const node_shape* shape, const node_params* params,
                       const float* y_traj, const float* grad_out,
                       const float* t_pts, int n_t,
                       float rtol, float atol, int method, const node_solve_opts* opts,
                       float* grad_y0, float* grad_params, float* grad_t,
                       node_stats* stats,
                       void* ws, size_t ws_bytes, void* stream);

/* Backward of torchdiffeq.odeint with adjoint=False (the constructor default, model.py:7; selected at model.py:359):
 * upstream backpropagates through the solver's operations.  The accepted steps of the forward solve are replayed
 * from y0 (`step_dt`: their sizes as node_solve_fwd logged them through node_solve_opts.dt_log, accepted ones only,
 * `n_steps` of them; ignored for rk4, whose grid is t_pts), all stage derivatives are kept, and the cotangent
 * grad_out [n_t, n, c, h, w] is carried back through every step: grad_y0 [n, c, h, w], grad_params flat in
 * parameters() order.  Step sizes are constants of the differentiation.  `rtol`, `atol`: the forward solve's -- they
 * select the convolution kernels exactly as node_solve_fwd did (the F(4x4,3x3) pipeline for dopri5 at rtol, atol >=
 * 1e-5 on 8x8 and 16x16 states), so the replayed stage values are the ones the forward output was computed from.  Workspace:
 * node_backprop_workspace_bytes (it holds the tape: 7 state-sized tensors per step). */
size_t node_backprop_workspace_bytes(const node_shape* shape, int method, int n_t, int n_steps);
int node_solve_backprop(const node_shape* shape, const node_params* params,
                        const float* y0, const float* t_pts, int n_t,
                        const double* step_dt, int n_steps, float rtol, float atol, int method,
                        const float* grad_out, float* grad_y0, float* grad_params,
                        void* ws, size_t ws_bytes, void* stream);

/* Classifier head behind the ODE block, up to (not including) the Linear layer
 *   -- model.py:231-250 (FCClassifier): GroupNorm(min(32,C), C) [model.py:268-271] -> ReLU ->
 *      AdaptiveAvgPool2d((1,1)) -> [Dropout] -> Flatten.
 * z: [n, c, h, w] (NCHW); gamma, beta: [c]; scale: [n, c] dropout mask / (1 - p) as the
 * caller's RNG produced it, or NULL (eval mode / no dropout); pooled: [n, c];
 * stats: [n, groups, 2] (mean, 1/sigma), kept by the caller for the backward.
 * No workspace; one launch each on `stream`. */
int node_head_fwd(const node_shape* shape, const float* z, const float* gamma, const float* beta,
                  const float* scale, float* pooled, float* stats, void* stream);

/* Backward of node_head_fwd: g_pooled [n, c] = dL/dpooled.  Outputs dz [n, c, h, w], the
 * per-sample partials gpart [n, 2, c] of (dL/dgamma, dL/dbeta) and -- gsum != NULL, a second small
 * launch -- their sums over n, gsum [2, c] = (dL/dgamma, dL/dbeta). */
int node_head_bwd(const node_shape* shape, const float* z, const float* gamma, const float* beta,
                  const float* scale, const float* stats, const float* g_pooled,
                  float* dz, float* gpart, float* gsum, void* stream);

/* GroupNorm (+ReLU) of the stem's residual blocks -- model.py:284-310 (`relu(norm(x))` in front of every
 * conv; normalization('group') = nn.GroupNorm(min(32,C), C), model.py:268-271), forward and backward.
 * z, out, g_out, dz: [n, c, h, w] (NCHW); stats: [n, groups, 2] (mean, 1/sigma) from the forward;
 * gpart: [n, 2, c] per-sample partials of (dL/dgamma, dL/dbeta); gsum: NULL, or [2, c] for their sums over n (a second
 * small launch).  relu != 0: out = relu(GN(z)); relu == 0: out = GN(z).  No workspace; one launch each. */
int node_gn_relu_fwd(const node_shape* shape, const float* z, const float* gamma, const float* beta,
                     int relu, float* out, float* stats, void* stream);
int node_gn_relu_bwd(const node_shape* shape, const float* z, const float* gamma, const float* beta,
                     const float* stats, int relu, const float* g_out, float* dz, float* gpart, float* gsum, void* stream);

/* Generic ("flat") solver -- the fallback of SURVEY.md 8b: `torchdiffeq.odeint[_adjoint]` accepts ANY nn.Module as `func`
 * (model.py:367), and the reference itself offers dynamics the fused kernels do not take (train.py:202 `--norm batch`).
 * For those the CALLER evaluates the dynamics (PyTorch ops on its stream) and this library does everything else on the
 * device with the kernels of the fused solves: stage states y + dt sum_j beta_ij k_j, stage times, the Hairer initial
 * step, the mixed error norm per tensor, accept / reject, the next step size, quartic dense output at the target times,
 * FSAL commit -- no host decision; the host reads the controller back when it likes (node_flat_status_read).
 * State: 1..3 flat fp32 tensors (`seg`), each with an end-of-step buffer y1 and seven stage-derivative buffers k[0..6]
 * owned by the caller, plus -- has_scalar -- one scalar kept inside the controller (the adjoint's time cotangent).
 * Time is in SOLVER ORIENTATION: increasing; a solve towards smaller t passes tsign = -1, negated times and derivatives
 * (upstream's convention), and node_flat_stage hands out tsign * time for the caller's function.
 * dopri5 step:  for s in 0..5: node_flat_stage(s) -> caller evaluates -> stores tsign * f into k[s + 1] (stage 5's state
 * must be written to the seg's y1 buffers: it IS y1); node_flat_finish_step.  Before the first step: stage NODE_FLAT_F0
 * (k[0]), then node_flat_initial_step(0), stage NODE_FLAT_PROBE (k[1]), node_flat_initial_step(1) unless first_dt is given.
 * rk4 (3/8 rule, one step per target interval): stage F0 -> k[0], stages 1..3 -> k[1..3], finish. */
typedef struct node_flat_seg {
  float* y;        /* state at the start of the current step (updated in place by the commit)   */
  float* y1;       /* state at the end of the step                                               */
  float* k[7];     /* stage derivatives                                                          */
  size_t n;
} node_flat_seg;
typedef struct node_flat_solve {
  int32_t nseg;          /* 1..3                                                                 */
  int32_t has_scalar;    /* != 0: augmented (adjoint) solve -- scalar segment, dense output of every segment at the interval's end */
  node_flat_seg seg[3];
  float rtol, atol, tsign;
  int32_t n_targets;     /* output times of the current interval                                 */
  void* ws;              /* node_flat_workspace_bytes(n_targets) bytes, 256-byte aligned          */
  size_t ws_bytes;
} node_flat_solve;
typedef struct node_flat_status {
  int32_t done, status, steps, accepted, rejected;
  double t, dt, first_dt;
  float scalar;
} node_flat_status;
enum { NODE_FLAT_F0 = -1, NODE_FLAT_PROBE = -2 };
size_t node_flat_workspace_bytes(int n_targets);
/* new interval starting at t0 (solver orientation) with `targets` (host array, increasing); first_dt = 0: to be chosen
 * by node_flat_initial_step.  new_solve != 0 also resets the cumulative counters and the scalar segment. */
int node_flat_begin(const node_flat_solve* f, double t0, const double* targets, double first_dt, int new_solve, void* stream);
/* y_stage[i] <- stage state of segment i (nothing for NODE_FLAT_F0); *t_stage (device float, nullable) <- tsign * stage time */
int node_flat_stage(const node_flat_solve* f, int method, int stage, float* const* y_stage, float* t_stage, void* stream);
/* scalar segment: which = -1 its value, 0..6 a stage derivative; (accumulate ? old : 0) + scale * src[0], src on the device */
int node_flat_scalar(const node_flat_solve* f, int which, const float* src, float scale, int accumulate, void* stream);
int node_flat_initial_step(const node_flat_solve* f, int phase, void* stream);
/* y_out (nullable, forward solves): [n_targets][seg[0].n], row j <- dense output at target j when a step passes it */
int node_flat_finish_step(const node_flat_solve* f, int method, float* y_out, void* stream);
int node_flat_status_read(const node_flat_solve* f, node_flat_status* out, void* stream);   /* synchronises the stream */

/* The classifier's Linear layer and the loss of the training loop as ONE launch each way -- model.py:244-250
 * (`nn.Linear(in_ch, out)` behind Flatten) and train.py:43 (`F.cross_entropy(p, y)`), plus the per-batch numbers the loop
 * reads at train.py:44,46 (loss value, correct predictions), left in device memory so that a caller can read them once
 * per logging interval instead of twice per batch.
 *   forward : logits[n][o] = sum_c pooled[n][c] weight[o][c] + bias[o]      (pooled == NULL: `logits` are given, not written)
 *             loss = mean_n | sum_n ( logsumexp_o logits[n] - logits[n][target[n]] )   (target == NULL: Linear alone)
 *             stat = { loss, #{n : argmax_o logits[n] == target[n]} }                   (optional)
 *   backward: d_logits = grad_loss * (softmax(logits) - onehot(target)) (/ n for the mean)   -- or `grad_logits` as given;
 *             d_weight[o][c] = sum_n d_logits[n][o] pooled[n][c], d_bias[o] = sum_n d_logits[n][o],
 *             d_pooled[n][c] = sum_o d_logits[n][o] weight[o][c]            (pooled == NULL: only `d_logits` is written)
 * fp32 throughout; class indices are int64 (what PyTorch hands over), every index in [0, classes) (no ignore_index:
 * an out-of-range target makes the loss NaN); classes <= 1024.  All sums in a fixed order: bit-reproducible.
 * `scratch`: node_head_loss_scratch_bytes(n) bytes of device memory, ZERO before its first use (the kernel leaves its
 * arrival counter at zero again), not shared between streams. */
enum { NODE_REDUCE_MEAN = 0, NODE_REDUCE_SUM = 1 };
typedef struct node_head_loss {
  int32_t n, c, classes;
  int32_t reduction;       /* NODE_REDUCE_MEAN (train.py:43) or NODE_REDUCE_SUM (train.py:93 test loss)            */
  const float* pooled;     /* [n, c]  what the head's pooling / dropout produced (node_head_fwd), or NULL          */
  const float* weight;     /* [classes, c]  nn.Linear.weight                                                      */
  const float* bias;       /* [classes] or NULL                                                                   */
  const int64_t* target;   /* [n] class indices, or NULL                                                          */
  float* logits;           /* [n, classes]                                                                        */
  float* loss;             /* [1]   (written when target != NULL)                                                 */
  float* stat;             /* [2] or NULL                                                                         */
  float* scratch;          /* node_head_loss_scratch_bytes(n)   (needed when target != NULL)                      */
} node_head_loss;
typedef struct node_head_loss_grad {
  const float* grad_loss;    /* [1] upstream gradient of the scalar loss (device), NULL = 1                       */
  const float* grad_logits;  /* [n, classes] given instead of the loss gradient (Linear alone), or NULL          */
  float* d_logits;           /* [n, classes] or NULL                                                              */
  float* d_pooled;           /* [n, c]                                                                            */
  float* d_weight;           /* [classes, c]                                                                      */
  float* d_bias;             /* [classes] or NULL                                                                 */
} node_head_loss_grad;
size_t node_head_loss_scratch_bytes(int n);
int node_head_loss_fwd(const node_head_loss* head, void* stream);
int node_head_loss_bwd(const node_head_loss* head, const node_head_loss_grad* grads, void* stream);

/* The optimizer step of the training loop -- train.py:136 (`torch.optim.SGD(params, lr, momentum=0.9, weight_decay=wd)`)
 * stepped at train.py:56-58 -- for ALL parameter tensors of the model in one launch (per 64 tensors):
 *   g = grad_scale * grad + weight_decay * p;  buf = momentum * buf + g;  p -= lr * buf
 * (dampening 0, no Nesterov: the reference's settings).  `tensors` is a HOST array of `count` records of device
 * pointers; gradients are read where autograd / the data-parallel reducer left them.  grad_scale folds the
 * 1/world of a data-parallel gradient SUM into the step.  momentum_buf must start at zero (the first step then
 * equals PyTorch's buf = g); it may be NULL when momentum == 0 (torch.optim.SGD keeps no buffer then).  `skip_if_nonzero` (device float, nullable): the launch leaves everything untouched
 * when it reads a non-zero value there -- the commit point of a step whose solves ran with deferred completion. */
typedef struct node_sgd_tensor {
  float* param;
  const float* grad;
  float* momentum_buf;
  size_t n;
} node_sgd_tensor;
int node_sgd_step(const node_sgd_tensor* tensors, int count, float lr, float momentum, float weight_decay,
                  float grad_scale, const float* skip_if_nonzero, void* stream);

/* The residual stem in front of the ODE block -- model.py:167-178 (`ResDownsample`):
 *     nn.Conv2d(in_ch, 64, 3, 1)                                              bias, no padding
 *     ResBlock(64, 64,      stride=2, downsample=conv1x1(64, 64, 2))          model.py:284-310
 *     ResBlock(64, filters, stride=2, downsample=conv1x1(64, filters, 2))
 * with ResBlock.forward (model.py:297-310): out = relu(norm1(x)); shortcut = downsample(out);
 * out = conv2(relu(norm2(conv1(out)))); return out + shortcut  -- conv3x3 / conv1x1 without bias (model.py:255-265),
 * norm = nn.GroupNorm(min(32, C), C) (model.py:268-271).  Forward and backward as hand-written gfx950 kernels, NHWC
 * inside the workspace: every convolution and data gradient on the bf16 matrix pipe at fp32 accuracy (activations,
 * gradients and filters as exact three-way bf16 splits, six products), weight gradients on the fp32 matrix
 * instructions.  x: [n, in_ch, h, w] NCHW (in_ch <= 3), out / grad_out: [n, filters, h2, w2] NCHW with
 * h0 = h - 2, h1 = (h0 - 1) / 2 + 1, h2 = (h1 - 1) / 2 + 1 (32 -> 30 -> 15 -> 8); filters % 64 == 0.
 * node_stem_bwd needs the workspace exactly as node_stem_fwd left it (the caller keeps it between the two calls);
 * every gradient pointer receives the full gradient (written, not accumulated). */
typedef struct node_stem_shape {
  int32_t n, in_ch, h, w, filters;
  float eps;
} node_stem_shape;
typedef struct node_stem_params {       /* state_dict keys under `downsample.module.` */
  const float* conv0_w;  /* 0.weight            [64, in_ch, 3, 3] */
  const float* conv0_b;  /* 0.bias              [64]              */
  const float* b1_n1_w;  /* 1.norm1.weight      [64]              */
  const float* b1_n1_b;  /* 1.norm1.bias        [64]              */
  const float* b1_c1_w;  /* 1.conv1.weight      [64, 64, 3, 3]    */
  const float* b1_n2_w;  /* 1.norm2.weight      [64]              */
  const float* b1_n2_b;  /* 1.norm2.bias        [64]              */
  const float* b1_c2_w;  /* 1.conv2.weight      [64, 64, 3, 3]    */
  const float* b1_ds_w;  /* 1.downsample.weight [64, 64, 1, 1]    */
  const float* b2_n1_w;  /* 2.norm1.weight      [64]              */
  const float* b2_n1_b;  /* 2.norm1.bias        [64]              */
  const float* b2_c1_w;  /* 2.conv1.weight      [filters, 64, 3, 3] */
  const float* b2_n2_w;  /* 2.norm2.weight      [filters]         */
  const float* b2_n2_b;  /* 2.norm2.bias        [filters]         */
  const float* b2_c2_w;  /* 2.conv2.weight      [filters, filters, 3, 3] */
  const float* b2_ds_w;  /* 2.downsample.weight [filters, 64, 1, 1] */
} node_stem_params;
typedef struct node_stem_grads {        /* same order and shapes as node_stem_params; device pointers, all required */
  float* conv0_w; float* conv0_b;
  float* b1_n1_w; float* b1_n1_b; float* b1_c1_w; float* b1_n2_w; float* b1_n2_b; float* b1_c2_w; float* b1_ds_w;
  float* b2_n1_w; float* b2_n1_b; float* b2_c1_w; float* b2_n2_w; float* b2_n2_b; float* b2_c2_w; float* b2_ds_w;
} node_stem_grads;
size_t node_stem_workspace_bytes(const node_stem_shape* shape);
int node_stem_fwd(const node_stem_shape* shape, const node_stem_params* params, const float* x, float* out,
                  void* ws, size_t ws_bytes, void* stream);
int node_stem_bwd(const node_stem_shape* shape, const node_stem_params* params, const float* x, const float* grad_out,
                  const node_stem_grads* grads, void* ws, size_t ws_bytes, void* stream);

/* Diagnostics (tests): ONE convolution of the stem's kernel family on NCHW fp32 tensors, through the same layout /
 * split / MFMA kernels node_stem_fwd and node_stem_bwd use.  what: 0 forward y = conv2d(x, w, stride, pad);
 * 1 data gradient dx = conv_transpose of dy (x_h, x_w: the input's spatial size); 2 weight gradient dw.
 * x: [n, cin, x_h, x_w]; w / dw: [cout, cin, k, k]; y / dy: [n, cout, y_h, y_w]; k in {1, 3}; cin, cout % 64 == 0. */
typedef struct node_conv_geom {
  int32_t n, cin, cout, x_h, x_w, k, stride, pad;
} node_conv_geom;
size_t node_stem_conv_workspace_bytes(const node_conv_geom* g);
int node_stem_conv(const node_conv_geom* g, int what, const float* x, const float* w, const float* dy, float* result,
                   void* ws, size_t ws_bytes, void* stream);

/* Event-based per-kernel-class timing (off by default; adds two event records
 * per profiled launch).  begin() resets the counters; end() synchronises the
 * recorded events and fills `out`. */
int node_profile_begin(void);
int node_profile_end(node_profile* out);

#ifdef __cplusplus
}
#endif
#endif /* NODE_HIP_H */
