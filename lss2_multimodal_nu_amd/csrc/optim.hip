// K10: the tail of a training step - total gradient norm, clip, Adam - over a LIST of fp32 tensors in three launches.
// replaces: torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm) + opt.step() with
//           opt = torch.optim.Adam(model.parameters(), lr, weight_decay=...) of the reference's loop (train.py:41, 62-63):
//           on the ATen path that is 17 launches per step here (per-tensor norms, stack, norm of norms, the clip
//           coefficient's five scalar kernels, two multi-tensor scalings, two multi-tensor Adam launches whose 64 Ki-element
//           chunks give the 4.65 M parameters of the model 71 workgroups: 1.3 TB/s).
//
//   launch 1  sum of squares: one workgroup per 4096-element chunk of one tensor -> partials[block] (fixed order)
//   launch 2  one workgroup: norm = sqrt(sum of partials in index order); coef = min(1, max_norm / (norm + 1e-6));
//             step += 1; the two bias corrections of that step
//   launch 3  per chunk: g *= coef (written back: `.grad` holds the clipped gradient afterwards, as after
//             clip_grad_norm_); g' = g + wd p; m = m + (g' - m)(1 - b1); v = b2 v + (1 - b2) g'^2;
//             p -= (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)          (torch.optim.Adam, amsgrad = False, maximize = False)
// The tensor list travels BY VALUE in the kernel arguments (40 B per tensor, up to 80 per launch): no device-side table
// to keep in step with gradient tensors that autograd re-allocates every eager step, and nothing a stream capture
// refuses.  All of it is bandwidth work: 28 B per parameter.
#include "lss_common.h"

namespace {

constexpr int OPT_MAX_T = 80;     // tensors per launch
constexpr int OPT_CHUNK = 4096;   // elements per workgroup: 256 threads x 4 x float4

struct OptTensor { float* p; float* g; float* m; float* v; long long n; };
struct OptTable {
  OptTensor t[OPT_MAX_T];
  int first[OPT_MAX_T + 1];   // first workgroup of tensor i; first[count] = workgroups of this launch
  int count;
};

// state[0] = step (float, counts completed steps), [1] = total norm before the clip, [2] = clip coefficient,
// [3] = 1 - b1^step, [4] = sqrt(1 - b2^step)
constexpr int ST_STEP = 0, ST_NORM = 1, ST_COEF = 2, ST_BC1 = 3, ST_BC2S = 4, ST_N = 8;

__device__ __forceinline__ int find_tensor(const OptTable& tab, int blk) {
  int lo = 0, hi = tab.count - 1;
  while (lo < hi) {  // last i with first[i] <= blk
    const int mid = (lo + hi + 1) >> 1;
    if (tab.first[mid] <= blk) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void sqnorm_partials_kernel(OptTable tab, float* __restrict__ partials, int block0) {
  __shared__ float red[4];
  const int i = find_tensor(tab, blockIdx.x);
  const OptTensor t = tab.t[i];
  const long long base = (long long)(blockIdx.x - tab.first[i]) * OPT_CHUNK;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long e = base + ((long long)k * 256 + threadIdx.x) * 4;
    if (e + 3 < t.n && (reinterpret_cast<uintptr_t>(t.g) & 15) == 0) {
      const float4 q = *reinterpret_cast<const float4*>(t.g + e);
      s = fmaf(q.x, q.x, s); s = fmaf(q.y, q.y, s); s = fmaf(q.z, q.z, s); s = fmaf(q.w, q.w, s);
    } else {
      for (int j = 0; j < 4; ++j)
        if (e + j < t.n) s = fmaf(t.g[e + j], t.g[e + j], s);
    }
  }
  s = lss_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[block0 + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void clip_finalize_kernel(const float* __restrict__ partials, int n, float max_norm,
                                                            float beta1, float beta2, float* __restrict__ state) {
  __shared__ float red[256];
  float s = 0.f;
  for (int k = threadIdx.x; k < n; k += 256) s += partials[k];   // thread t: partials t, t + 256, ... in order
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {   // fixed tree
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = sqrtf(red[0]);
    state[ST_NORM] = norm;
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.f);
    state[ST_COEF] = coef;
    const float step = state[ST_STEP] + 1.f;
    state[ST_STEP] = step;
    state[ST_BC1] = 1.f - powf(beta1, step);
    state[ST_BC2S] = sqrtf(1.f - powf(beta2, step));
  }
}

__global__ __launch_bounds__(256) void clip_adam_kernel(OptTable tab, const float* __restrict__ state, float lr,
                                                        float beta1, float beta2, float eps, float wd) {
  const int i = find_tensor(tab, blockIdx.x);
  const OptTensor t = tab.t[i];
  const long long base = (long long)(blockIdx.x - tab.first[i]) * OPT_CHUNK;
  const float coef = state[ST_COEF];
  const float step_size = lr / state[ST_BC1];
  const float bc2s = state[ST_BC2S];
  const bool vec = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) | reinterpret_cast<uintptr_t>(t.m) |
                     reinterpret_cast<uintptr_t>(t.v)) & 15) == 0;
  auto one = [&](float& p, float& g, float& m, float& v) {
    g *= coef;
    const float gg = wd != 0.f ? fmaf(wd, p, g) : g;
    m = fmaf(gg - m, 1.f - beta1, m);
    v = fmaf(beta2, v, (1.f - beta2) * gg * gg);
    p -= step_size * m / (sqrtf(v) / bc2s + eps);
  };
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long e = base + ((long long)k * 256 + threadIdx.x) * 4;
    if (e >= t.n) continue;
    if (vec && e + 3 < t.n) {
      float4 p = *reinterpret_cast<float4*>(t.p + e), g = *reinterpret_cast<float4*>(t.g + e);
      float4 m = *reinterpret_cast<float4*>(t.m + e), v = *reinterpret_cast<float4*>(t.v + e);
      one(p.x, g.x, m.x, v.x); one(p.y, g.y, m.y, v.y); one(p.z, g.z, m.z, v.z); one(p.w, g.w, m.w, v.w);
      *reinterpret_cast<float4*>(t.p + e) = p; *reinterpret_cast<float4*>(t.g + e) = g;
      *reinterpret_cast<float4*>(t.m + e) = m; *reinterpret_cast<float4*>(t.v + e) = v;
    } else {
      for (int j = 0; j < 4 && e + j < t.n; ++j) one(t.p[e + j], t.g[e + j], t.m[e + j], t.v[e + j]);
    }
  }
}

}  // namespace

// workgroups (= floats of `partials`) the step over tensors of these sizes needs
extern "C" long long lss_clip_adam_partials(const long long* numel, int count) {
  if (numel == nullptr || count <= 0) return 0;
  long long nb = 0;
  for (int i = 0; i < count; ++i) {
    if (numel[i] <= 0) return 0;
    nb += (numel[i] + OPT_CHUNK - 1) / OPT_CHUNK;
  }
  return nb;
}

// One clip + Adam step.  tensors: HOST array of `count` entries {p, g, m, v (device, fp32, contiguous), n}; state:
// 8 device floats, zero before the first step (state[0] counts the steps, state[1] = the total gradient norm before
// the clip, state[2] = the clip coefficient of this step); partials: lss_clip_adam_partials(...) device floats.
// max_norm <= 0: no clip.  Everything is launched on `stream`; nothing is read back.
extern "C" int lss_clip_adam_step(const void* tensors, int count, float* state, float* partials, long long n_partials,
                                  float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                                  void* stream) {
  LSS_CHECK_PTR(tensors); LSS_CHECK_PTR(state); LSS_CHECK_PTR(partials);
  if (count <= 0 || !(lr >= 0.f) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f))
    return LSS_E_SHAPE;
  const OptTensor* in = static_cast<const OptTensor*>(tensors);
  long long nb = 0;
  for (int i = 0; i < count; ++i) {
    if (in[i].p == nullptr || in[i].g == nullptr || in[i].m == nullptr || in[i].v == nullptr) return LSS_E_NULL;
    if (in[i].n <= 0 || in[i].n >= (1LL << 40)) return LSS_E_SHAPE;
    if (((reinterpret_cast<uintptr_t>(in[i].p) | reinterpret_cast<uintptr_t>(in[i].g) | reinterpret_cast<uintptr_t>(in[i].m) |
          reinterpret_cast<uintptr_t>(in[i].v)) & 3) != 0)
      return LSS_E_ALIGN;
    nb += (in[i].n + OPT_CHUNK - 1) / OPT_CHUNK;
  }
  if (nb != n_partials || nb >= (1LL << 30)) return LSS_E_SHAPE;
  hipStream_t st = lss_stream(stream);
  for (int pass = 0; pass < 2; ++pass) {
    int block0 = 0;
    for (int i0 = 0; i0 < count; i0 += OPT_MAX_T) {
      OptTable tab;
      tab.count = count - i0 < OPT_MAX_T ? count - i0 : OPT_MAX_T;
      int b = 0;
      for (int i = 0; i < tab.count; ++i) {
        tab.t[i] = in[i0 + i];
        tab.first[i] = b;
        b += (int)((in[i0 + i].n + OPT_CHUNK - 1) / OPT_CHUNK);
      }
      tab.first[tab.count] = b;
      if (pass == 0)
        hipLaunchKernelGGL(sqnorm_partials_kernel, dim3(b), dim3(256), 0, st, tab, partials, block0);
      else
        hipLaunchKernelGGL(clip_adam_kernel, dim3(b), dim3(256), 0, st, tab, state, lr, beta1, beta2, eps, weight_decay);
      block0 += b;
    }
    if (pass == 0)
      hipLaunchKernelGGL(clip_finalize_kernel, dim3(1), dim3(256), 0, st, partials, (int)nb, max_norm, beta1, beta2,
                         state);
  }
  return lss_launch_status();
}
