// AdaRound / BRECQ / QDrop inner loop on MI355X: the elementwise work around the layer's
// convolution, fused so that one learning iteration is  conv forward -> dpl_l2_loss -> conv backward ->
// dpl_round_step  (the reference runs ~60 eager torch kernels per iteration for the same arithmetic:
// weight_transform/ada_quant_layer.py:28-50,96-125, adaround.py:119-144, torch.optim.Adam).
//
//   k_round_init   floor(w / scale) and the initial round mask                     ada_quant_layer.py:147
//   k_round_quant  soft / hard quantised weight                                    ada_quant_layer.py:39-50
//   k_round_step   dL/d(mask) from dL/d(qw) through the soft quantiser + the rounding regulariser's value and
//                  gradient + one Adam update + the next iteration's soft-quantised weight, in ONE pass over
//                  the weight-sized arrays
//   k_l2_loss      L2_norm(relu?(z), target) and its gradient w.r.t. z in one read of both tensors
//   k_acti_drop_*  quant_acti with QDrop mixing and its (reference-defined) gradient
//   k_sparse_*     sparse + quantised weight with straight-through rounding, its gradient fused with the SGD update
//                  (sparse_quant_layer.py:9-66, sparse_quant.py:107-109)
//
// All of it is launch- and HBM-bound elementwise work: 16 B per lane where the layout allows, fp32 arithmetic in
// the reference's operation order (IEEE divide, no FMA contraction), fp64 block-reduced loss sums.
#include "common.hpp"

#pragma clang fp contract(off)

namespace {

constexpr float kZetaMinusGamma = 1.2f;   // fp32(1.1 - (-0.1)) as torch casts the python scalar
constexpr float kGamma = -0.1f;

struct RectSig {
    float h;    // clamp((zeta - gamma) * sigmoid(a) + gamma, 0, 1)
    float dh;   // dh / da (0 where the clamp is active)
};

__device__ __forceinline__ RectSig rect_sigmoid(float a) {
    const float sg = __fdiv_rn(1.0f, 1.0f + expf(-a));
    const float hr = kZetaMinusGamma * sg + kGamma;
    RectSig r;
    r.h = fminf(fmaxf(hr, 0.0f), 1.0f);
    r.dh = (hr >= 0.0f && hr <= 1.0f) ? (kZetaMinusGamma * (1.0f - sg)) * sg : 0.0f;
    return r;
}

// torch.maximum / minimum backward: the full gradient to the larger (smaller) side, half of it on a tie.
__device__ __forceinline__ float clamp_pass(float v0, float qmin, float qmax, float& v_out) {
    const float f_lo = v0 > qmin ? 1.0f : (v0 == qmin ? 0.5f : 0.0f);
    const float v1 = fmaxf(v0, qmin);
    const float f_hi = v1 < qmax ? 1.0f : (v1 == qmax ? 0.5f : 0.0f);
    v_out = fminf(v1, qmax);
    return f_lo * f_hi;
}

struct ChannelParams {
    const float* scale;
    const float* qmin;
    const float* qmax;
    uint32_t n_channels, inner;
    __device__ __forceinline__ uint32_t channel(uint32_t idx) const { return n_channels > 1 ? idx / inner : 0u; }
};

__global__ __launch_bounds__(kBlock) void k_round_init(const float* __restrict__ w, ChannelParams cp, uint32_t n,
                                                        float* __restrict__ wfloor, float* __restrict__ alpha) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float t = __fdiv_rn(w[i], cp.scale[cp.channel(i)]);
        const float wf = floorf(t);
        const float rest = t - wf;
        wfloor[i] = wf;
        // torch evaluates scalar / tensor as reciprocal(tensor) * scalar
        alpha[i] = -logf(__fdiv_rn(1.0f, rest - kGamma) * kZetaMinusGamma - 1.0f);
    }
}

__global__ __launch_bounds__(kBlock) void k_round_quant(const float* __restrict__ wfloor,
                                                         const float* __restrict__ alpha, ChannelParams cp,
                                                         uint32_t n, int clamp, int soft, float* __restrict__ qw) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const uint32_t c = cp.channel(i);
        const float a = alpha[i];
        const float h = soft ? rect_sigmoid(a).h : (a >= 0.0f ? 1.0f : 0.0f);
        float v = wfloor[i] + h;
        if (clamp) clamp_pass(v, cp.qmin[c], cp.qmax[c], v);
        qw[i] = v * cp.scale[c];
    }
}

struct StepParams {
    float grad_scale;   // multiplies dL/d(qw) (1 / world_size after a SUM all-reduce: DDP's mean)
    float beta;         // regulariser temperature (0: the regulariser and its gradient are zero)
    float lambda;       // regulariser weight (0.01)
    float step_size;    // lr / (1 - beta1^t)
    float bc2_sqrt;     // sqrt(1 - beta2^t)
    float one_minus_beta1, beta2, one_minus_beta2, eps;   // python doubles cast to fp32, as torch passes them
    int clamp;          // per-channel clamp of floor + h (the reference's per-tensor branch does not clamp)
    int adam;           // 0: gradients only (alpha, m, v, qw untouched)
};

__global__ __launch_bounds__(kBlock) void k_round_step(const float* __restrict__ grad_qw,
                                                        const float* __restrict__ wfloor, float* __restrict__ alpha,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        ChannelParams cp, uint32_t n, StepParams sp,
                                                        const dpl_round_sched* __restrict__ sched,
                                                        float* __restrict__ qw_next, float* __restrict__ grad_alpha,
                                                        double* __restrict__ reg_loss) {
    __shared__ double s_red[kBlock / kWave];
    if (sched) {  // captured in a hipGraph: this iteration's temperature and Adam corrections live on the device
        sp.beta = sched->reg_beta;
        sp.step_size = sched->step_size;
        sp.bc2_sqrt = sched->bc2_sqrt;
    }
    double reg_part = 0.0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const uint32_t c = cp.channel(i);
        const float s = cp.scale[c];
        const float a = alpha[i];
        const float wf = wfloor[i];
        const RectSig r = rect_sigmoid(a);
        float vq = wf + r.h;
        float pass = 1.0f;
        if (sp.clamp) pass = clamp_pass(vq, cp.qmin[c], cp.qmax[c], vq);
        // dL/d(mask): (dL/dqw * scale) through the clamp, the add and the rectified sigmoid
        float g = grad_qw ? (((grad_qw[i] * sp.grad_scale) * s) * pass) * r.dh : 0.0f;
        if (sp.beta > 0.0f) {  // lambda * sum(1 - (|h - 0.5| * 2)^beta)
            const float d = r.h - 0.5f;
            const float u = fabsf(d) * 2.0f;
            reg_part += (double)(1.0f - powf(u, sp.beta));
            const float dp = u > 0.0f ? sp.beta * powf(u, sp.beta - 1.0f) : 0.0f;          // d(u^beta)/du
            const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
            g += ((-sp.lambda * dp) * 2.0f * sgn) * r.dh;
        }
        if (grad_alpha) grad_alpha[i] = g;
        if (sp.adam) {  // torch.optim.Adam (single-tensor form): lerp, addcmul, addcdiv
            const float mi = m[i] + sp.one_minus_beta1 * (g - m[i]);
            const float vi = v[i] * sp.beta2 + (sp.one_minus_beta2 * g) * g;
            const float denom = __fdiv_rn(sqrtf(vi), sp.bc2_sqrt) + sp.eps;
            const float an = a + __fdiv_rn((-sp.step_size) * mi, denom);
            m[i] = mi;
            v[i] = vi;
            alpha[i] = an;
            if (qw_next) {
                float vn = wf + rect_sigmoid(an).h;
                if (sp.clamp) clamp_pass(vn, cp.qmin[c], cp.qmax[c], vn);
                qw_next[i] = vn * s;
            }
        }
    }
    if (reg_loss && sp.beta > 0.0f) {
        reg_part = wave_sum(reg_part);
        if ((threadIdx.x & (kWave - 1)) == 0) s_red[threadIdx.x / kWave] = reg_part;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int j = 0; j < kBlock / kWave; ++j) t += s_red[j];
            atomicAdd(reg_loss, (double)sp.lambda * t);
        }
    }
}

// One thread: advance the learner's schedule by one iteration (TempDecay, ada_quant_layer.py:119-134, and Adam's
// bias corrections as torch computes them on the host, in double).
__global__ void k_round_sched_advance(dpl_round_sched* __restrict__ sc, int32_t t_max, double lr, double beta1,
                                      double beta2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int32_t t = sc->iter;
    const double start = 0.2 * (double)t_max;
    double beta = 0.0;
    if (!((double)t < start)) {
        const double rel_t = ((double)t - start) / ((double)t_max - start);
        beta = 2.0 + 0.5 * (20.0 - 2.0) * (1.0 + cos(rel_t * 3.141592653589793));
    }
    const int32_t step = sc->adam_step + 1;
    sc->reg_beta = (float)beta;
    sc->step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
    sc->bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    sc->adam_step = step;
    sc->iter = t + 1;
}

// ---------------------------------------------------------------- sparse + quantised weights (sparse_quant_layer.py)
// qw = clamp?(rint(w * mask / scale)) * scale with a straight-through round (STE, :9-18): forward, and the fused
// "backward + SGD" update  g = ((dL/dqw * gs * scale) * pass) / scale * mask;  g += wd * w;
// buf = first ? g : momentum * buf + g;  w -= lr * buf   (torch.optim.SGD, single-tensor form).
__global__ __launch_bounds__(kBlock) void k_sparse_quant(const float* __restrict__ w, const float* __restrict__ mask,
                                                          ChannelParams cp, uint32_t n, int clamp,
                                                          float* __restrict__ qw) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const uint32_t c = cp.channel(i);
        const float s = cp.scale[c];
        float v = rintf(__fdiv_rn(mask ? w[i] * mask[i] : w[i], s));
        if (clamp) clamp_pass(v, cp.qmin[c], cp.qmax[c], v);
        qw[i] = v * s;
    }
}

struct SgdParams {
    float grad_scale, lr, momentum, weight_decay;
    int clamp, first, update;
};

__global__ __launch_bounds__(kBlock) void k_sparse_step(const float* __restrict__ grad_qw, float* __restrict__ w,
                                                         const float* __restrict__ mask, float* __restrict__ buf,
                                                         ChannelParams cp, uint32_t n, SgdParams sp,
                                                         float* __restrict__ grad_w) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const uint32_t c = cp.channel(i);
        const float s = cp.scale[c];
        const float wi = w[i];
        const float mk = mask ? mask[i] : 1.0f;
        float pass = 1.0f;
        if (sp.clamp) {
            float v = rintf(__fdiv_rn(wi * mk, s));
            pass = clamp_pass(v, cp.qmin[c], cp.qmax[c], v);
        }
        float g = __fdiv_rn(((grad_qw[i] * sp.grad_scale) * s) * pass, s) * mk;
        if (grad_w) grad_w[i] = g;
        if (sp.update) {
            if (sp.weight_decay != 0.0f) g = g + sp.weight_decay * wi;
            const float b = sp.first ? g : buf[i] * sp.momentum + g;
            buf[i] = b;
            w[i] = wi + (-sp.lr) * b;
        }
    }
}

// loss += sum((relu?(z) - t)^2) * inv_m ;  grad = coef * (relu?(z) - t) * (z > 0 if relu)
template <bool kVec>
__global__ __launch_bounds__(kBlock) void k_l2_loss(const float* __restrict__ z, const float* __restrict__ t,
                                                     uint64_t n, int relu, float coef, double inv_m,
                                                     float* __restrict__ grad, double* __restrict__ loss) {
    __shared__ double s_red[kBlock / kWave];
    double part = 0.0;
    auto one = [&](float zi, float ti) -> float {
        const float y = relu ? fmaxf(zi, 0.0f) : zi;
        const float d = y - ti;
        part += (double)(d * d);
        return (relu && !(zi > 0.0f)) ? 0.0f : coef * d;
    };
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    if (kVec) {
        const uint64_t nv = n >> 2;
        const f4* zv = reinterpret_cast<const f4*>(z);
        const f4* tv = reinterpret_cast<const f4*>(t);
        f4* gv = reinterpret_cast<f4*>(grad);
        for (uint64_t i0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i0 < nv; i0 += 4 * stride) {
            f4 a[4], b[4];   // eight 16-byte loads in flight per lane
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint64_t i = i0 + u * stride;
                a[u] = i < nv ? __builtin_nontemporal_load(zv + i) : f4{0.f, 0.f, 0.f, 0.f};
                b[u] = i < nv ? __builtin_nontemporal_load(tv + i) : f4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint64_t i = i0 + u * stride;
                if (i < nv) {
                    f4 g;
                    g.x = one(a[u].x, b[u].x);
                    g.y = one(a[u].y, b[u].y);
                    g.z = one(a[u].z, b[u].z);
                    g.w = one(a[u].w, b[u].w);
                    if (grad) __builtin_nontemporal_store(g, gv + i);
                }
            }
        }
        for (uint64_t i = (nv << 2) + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
            const float g = one(z[i], t[i]);
            if (grad) grad[i] = g;
        }
    } else {
        for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
            const float g = one(z[i], t[i]);
            if (grad) grad[i] = g;
        }
    }
    part = wave_sum(part);
    if ((threadIdx.x & (kWave - 1)) == 0) s_red[threadIdx.x / kWave] = part;
    __syncthreads();
    if (threadIdx.x == 0 && loss) {
        double tt = 0.0;
        for (int j = 0; j < kBlock / kWave; ++j) tt += s_red[j];
        atomicAdd(loss, tt * inv_m);
    }
}

// quant_acti (ada_quant_layer.py:28-36): y = r < prob ? clamp(rint(x / s), lo, hi) * s : x.  Its autograd gradient
// is zero through round() and 1 through the untouched branch.
__global__ __launch_bounds__(kBlock) void k_acti_drop_fwd(const float* __restrict__ x, const float* __restrict__ r,
                                                           uint64_t n, float scale, float qmin, float qmax,
                                                           float prob, float* __restrict__ y) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const float xi = x[i];
        float q = rintf(__fdiv_rn(xi, scale));
        q = fminf(fmaxf(q, qmin), qmax) * scale;
        y[i] = (!r || r[i] < prob) ? q : xi;
    }
}

__global__ __launch_bounds__(kBlock) void k_acti_drop_bwd(const float* __restrict__ r, const float* __restrict__ gy,
                                                           uint64_t n, float prob, float* __restrict__ gx) {
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        gx[i] = (!r || r[i] < prob) ? 0.0f : gy[i];
}

inline unsigned blocks_for(uint64_t n, int per_thread) {
    uint64_t b = (n / per_thread + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    if (b > 256 * 16) b = 256 * 16;
    return (unsigned)b;
}

int check_channels(const char* who, int64_t n, int64_t n_channels, int64_t inner) {
    if (n <= 0 || n > 0xFFFFFFFFll) {
        snprintf(g_err, sizeof(g_err), "%s: n must be in [1, 2^32)", who);
        return -2;
    }
    if (n_channels < 1 || inner < 1 || (n_channels > 1 && n_channels * inner != n)) {
        snprintf(g_err, sizeof(g_err), "%s: n must equal n_channels * inner", who);
        return -2;
    }
    return 0;
}

}  // namespace

extern "C" {

int dpl_round_init(const float* d_w, const float* d_scale, int64_t n, int64_t n_channels, int64_t inner,
                   float* d_wfloor, float* d_alpha, dpl_stream_t s) {
    if (int e = check_channels("dpl_round_init", n, n_channels, inner)) return e;
    const ChannelParams cp{d_scale, nullptr, nullptr, (uint32_t)n_channels, (uint32_t)inner};
    hipLaunchKernelGGL(k_round_init, dim3(blocks_for((uint64_t)n, 1)), dim3(kBlock), 0, (hipStream_t)s, d_w, cp,
                       (uint32_t)n, d_wfloor, d_alpha);
    DPL_LAUNCH_CHECK("k_round_init");
    return 0;
}

int dpl_round_quant(const float* d_wfloor, const float* d_alpha, const float* d_scale, const float* d_qmin,
                    const float* d_qmax, int64_t n, int64_t n_channels, int64_t inner, int clamp, int soft,
                    float* d_qw, dpl_stream_t s) {
    if (int e = check_channels("dpl_round_quant", n, n_channels, inner)) return e;
    if (clamp && (!d_qmin || !d_qmax)) return fail_msg("dpl_round_quant: clamp needs q_min and q_max");
    const ChannelParams cp{d_scale, d_qmin, d_qmax, (uint32_t)n_channels, (uint32_t)inner};
    hipLaunchKernelGGL(k_round_quant, dim3(blocks_for((uint64_t)n, 1)), dim3(kBlock), 0, (hipStream_t)s, d_wfloor,
                       d_alpha, cp, (uint32_t)n, clamp, soft, d_qw);
    DPL_LAUNCH_CHECK("k_round_quant");
    return 0;
}

int dpl_round_step(const float* d_grad_qw, const float* d_wfloor, float* d_alpha, float* d_m, float* d_v,
                   const float* d_scale, const float* d_qmin, const float* d_qmax, int64_t n, int64_t n_channels,
                   int64_t inner, const dpl_round_step_params* p, const dpl_round_sched* d_sched, float* d_qw_next,
                   float* d_grad_alpha, double* d_reg_loss, dpl_stream_t s) {
    if (int e = check_channels("dpl_round_step", n, n_channels, inner)) return e;
    if (!p) return fail_msg("dpl_round_step: params missing");
    if (p->clamp && (!d_qmin || !d_qmax)) return fail_msg("dpl_round_step: clamp needs q_min and q_max");
    if (p->adam && (!d_m || !d_v || (p->step < 1 && !d_sched)))
        return fail_msg("dpl_round_step: Adam needs moments and step >= 1 (or a device schedule)");
    StepParams sp;
    sp.grad_scale = p->grad_scale;
    sp.beta = p->reg_beta;
    sp.lambda = p->reg_lambda;
    // bias corrections as torch computes them: in double on the host, then cast
    const int step = p->step < 1 ? 1 : p->step;
    const double bc1 = 1.0 - pow(p->adam_beta1, (double)step);
    const double bc2 = 1.0 - pow(p->adam_beta2, (double)step);
    sp.step_size = p->adam ? (float)(p->lr / bc1) : 0.0f;
    sp.bc2_sqrt = p->adam ? (float)sqrt(bc2) : 1.0f;
    sp.one_minus_beta1 = (float)(1.0 - p->adam_beta1);
    sp.beta2 = (float)p->adam_beta2;
    sp.one_minus_beta2 = (float)(1.0 - p->adam_beta2);
    sp.eps = (float)p->adam_eps;
    sp.clamp = p->clamp;
    sp.adam = p->adam;
    const ChannelParams cp{d_scale, d_qmin, d_qmax, (uint32_t)n_channels, (uint32_t)inner};
    hipLaunchKernelGGL(k_round_step, dim3(blocks_for((uint64_t)n, 1)), dim3(kBlock), 0, (hipStream_t)s, d_grad_qw,
                       d_wfloor, d_alpha, d_m, d_v, cp, (uint32_t)n, sp, d_sched, d_qw_next, d_grad_alpha, d_reg_loss);
    DPL_LAUNCH_CHECK("k_round_step");
    return 0;
}

int dpl_sparse_quant(const float* d_w, const float* d_mask, const float* d_scale, const float* d_qmin,
                     const float* d_qmax, int64_t n, int64_t n_channels, int64_t inner, int clamp, float* d_qw,
                     dpl_stream_t s) {
    if (int e = check_channels("dpl_sparse_quant", n, n_channels, inner)) return e;
    if (clamp && (!d_qmin || !d_qmax)) return fail_msg("dpl_sparse_quant: clamp needs q_min and q_max");
    const ChannelParams cp{d_scale, d_qmin, d_qmax, (uint32_t)n_channels, (uint32_t)inner};
    hipLaunchKernelGGL(k_sparse_quant, dim3(blocks_for((uint64_t)n, 1)), dim3(kBlock), 0, (hipStream_t)s, d_w, d_mask,
                       cp, (uint32_t)n, clamp, d_qw);
    DPL_LAUNCH_CHECK("k_sparse_quant");
    return 0;
}

int dpl_sparse_step(const float* d_grad_qw, float* d_w, const float* d_mask, float* d_momentum_buf,
                    const float* d_scale, const float* d_qmin, const float* d_qmax, int64_t n, int64_t n_channels,
                    int64_t inner, int clamp, float grad_scale, float lr, float momentum, float weight_decay, int first,
                    int update, float* d_grad_w, dpl_stream_t s) {
    if (int e = check_channels("dpl_sparse_step", n, n_channels, inner)) return e;
    if (!d_grad_qw) return fail_msg("dpl_sparse_step: gradient missing");
    if (clamp && (!d_qmin || !d_qmax)) return fail_msg("dpl_sparse_step: clamp needs q_min and q_max");
    if (update && !d_momentum_buf) return fail_msg("dpl_sparse_step: the update needs a momentum buffer");
    const ChannelParams cp{d_scale, d_qmin, d_qmax, (uint32_t)n_channels, (uint32_t)inner};
    const SgdParams sp{grad_scale, lr, momentum, weight_decay, clamp, first, update};
    hipLaunchKernelGGL(k_sparse_step, dim3(blocks_for((uint64_t)n, 1)), dim3(kBlock), 0, (hipStream_t)s, d_grad_qw, d_w,
                       d_mask, d_momentum_buf, cp, (uint32_t)n, sp, d_grad_w);
    DPL_LAUNCH_CHECK("k_sparse_step");
    return 0;
}

int dpl_round_sched_advance(dpl_round_sched* d_sched, int32_t t_max, double lr, double adam_beta1, double adam_beta2,
                            dpl_stream_t s) {
    if (!d_sched || t_max < 1) return fail_msg("dpl_round_sched_advance: bad arguments");
    hipLaunchKernelGGL(k_round_sched_advance, dim3(1), dim3(1), 0, (hipStream_t)s, d_sched, t_max, lr, adam_beta1,
                       adam_beta2);
    DPL_LAUNCH_CHECK("k_round_sched_advance");
    return 0;
}

int dpl_l2_loss(const float* d_z, const float* d_target, int64_t n, int relu, float grad_coef, double inv_m,
                float* d_grad, double* d_loss, dpl_stream_t s) {
    if (n <= 0) return 0;
    const bool vec = ((((uintptr_t)d_z | (uintptr_t)d_target | (uintptr_t)d_grad) & 15u) == 0);
    const dim3 g(blocks_for((uint64_t)n, 8)), b(kBlock);
    if (vec)
        hipLaunchKernelGGL(k_l2_loss<true>, g, b, 0, (hipStream_t)s, d_z, d_target, (uint64_t)n, relu, grad_coef,
                           inv_m, d_grad, d_loss);
    else
        hipLaunchKernelGGL(k_l2_loss<false>, g, b, 0, (hipStream_t)s, d_z, d_target, (uint64_t)n, relu, grad_coef,
                           inv_m, d_grad, d_loss);
    DPL_LAUNCH_CHECK("k_l2_loss");
    return 0;
}

int dpl_acti_drop_fwd(const float* d_x, const float* d_rand, int64_t n, float scale, float qmin, float qmax,
                      float prob, float* d_y, dpl_stream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_acti_drop_fwd, dim3(blocks_for((uint64_t)n, 4)), dim3(kBlock), 0, (hipStream_t)s, d_x,
                       d_rand, (uint64_t)n, scale, qmin, qmax, prob, d_y);
    DPL_LAUNCH_CHECK("k_acti_drop_fwd");
    return 0;
}

int dpl_acti_drop_bwd(const float* d_rand, const float* d_grad_y, int64_t n, float prob, float* d_grad_x,
                      dpl_stream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_acti_drop_bwd, dim3(blocks_for((uint64_t)n, 4)), dim3(kBlock), 0, (hipStream_t)s, d_rand,
                       d_grad_y, (uint64_t)n, prob, d_grad_x);
    DPL_LAUNCH_CHECK("k_acti_drop_bwd");
    return 0;
}

}  // extern "C"
