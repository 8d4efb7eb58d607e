// Dropout / DropPath multipliers from a counter-based generator (Philox4x32-10, Salmon et al. 2011) -- replaces the framework's
// fused_dropout / bernoulli launches of the training step (reference sites: torchscale DropPath at beit3_base.py:146-151,
// nn.Dropout / attention dropout of the DETR decoder layers at heads/tgqs_kd_detr_head/transformer.py:106-125; same
// distribution, not the same stream, like any two seeds of the reference).  One launch fills ALL multipliers a step needs
// (the head draws one buffer per rate, the encoder one [L, 2, B] table): element i is a pure function of (seed, offset, i), so a
// buffer can be regenerated bit for bit.  Eagerly launched steps keep no generator state on the device (the host passes a fresh
// key per call); a launch recorded into a hipGraph has its arguments frozen, so it takes a device-side epoch instead: the
// kernel mixes *epoch into the counter's high words and the last workgroup to finish increments it, so every replay of the
// graph draws new multipliers.
#include "common.h"

namespace {

struct PhiloxOut { unsigned v[4]; };

__host__ __device__ inline PhiloxOut philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  PhiloxOut o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// out[i] = u_i < keep ? 1 / keep : 0,  u_i = 24 high bits of word (i & 3) of Philox(counter = {(i >> 2) + offset, epoch}, key = seed);
// keep = keep_seg[i / seg] when given (DropPath: one keep probability per layer), else the scalar
__global__ __launch_bounds__(256) void dropout_mult_kernel(float* __restrict__ out, long n, float keep, const float* __restrict__ keep_seg,
                                                           long seg, unsigned long long seed, unsigned long long offset,
                                                           unsigned long long* __restrict__ state) {
  const long quads = (n + 3) >> 2;
  const unsigned long long epoch = state ? __atomic_load_n(state, __ATOMIC_RELAXED) : 0ull;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (long)gridDim.x * blockDim.x) {
    const unsigned long long ctr = (unsigned long long)q + offset;
    const PhiloxOut r = philox4x32_10((unsigned)ctr, (unsigned)(ctr >> 32), (unsigned)epoch, (unsigned)(epoch >> 32), (unsigned)seed, (unsigned)(seed >> 32));
    float m[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long i = q * 4 + e;
      const float kp = keep_seg ? keep_seg[(i < n ? i : n - 1) / seg] : keep;
      const float u = (float)(r.v[e] >> 8) * (1.0f / 16777216.0f);
      m[e] = u < kp ? 1.0f / kp : 0.0f;
    }
    if (q * 4 + 3 < n) *(f32x4_t*)(out + q * 4) = (f32x4_t){m[0], m[1], m[2], m[3]};
    else for (int e = 0; e < 4 && q * 4 + e < n; ++e) out[q * 4 + e] = m[e];
  }
  if (state) {      // state[0] = epoch, state[1] = ticket: every workgroup has read the epoch before it takes its ticket
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long t = atomicAdd(state + 1, 1ull);
      if (t == (unsigned long long)gridDim.x - 1) {
        __atomic_store_n(state + 1, 0ull, __ATOMIC_RELAXED);
        __atomic_store_n(state, epoch + 1, __ATOMIC_RELAXED);
      }
    }
  }
}

}  // namespace

// host-side evaluation of the generator (known-answer tests without a GPU): out[4] = Philox4x32-10(ctr[4], key[2])
extern "C" int simvg_philox4x32(const unsigned* ctr, const unsigned* key, unsigned* out) {
  SIMVG_CHECK_ARG(ctr && key && out, "philox4x32: null argument");
  const PhiloxOut r = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
  for (int e = 0; e < 4; ++e) out[e] = r.v[e];
  return SIMVG_OK;
}

extern "C" int simvg_dropout_mult(float* out, long n, float keep, const float* keep_seg, long seg, unsigned long long seed,
                                  unsigned long long offset, unsigned long long* state, hipStream_t stream) {
  SIMVG_CHECK_ARG(out && n > 0 && ((uintptr_t)out & 15) == 0, "dropout_mult: need a 16-B aligned output and n > 0");
  SIMVG_CHECK_ARG(keep_seg ? seg > 0 : (keep > 0.f && keep <= 1.f), "dropout_mult: keep probability in (0, 1] (or per-segment table + segment length)");
  const long quads = (n + 3) / 4;
  const int blocks = (int)std::min<long>((quads + 255) / 256, 2048);
  hipLaunchKernelGGL(dropout_mult_kernel, dim3(blocks), dim3(256), 0, stream, out, n, keep, keep_seg, seg, seed, offset, state);
  SIMVG_LAUNCH_CHECK();
  return SIMVG_OK;
}
