// Adam over many tensors in ONE launch (the optimizer of both networks in the reference's training step: torch.optim.Adam,
// codes/models/SRRaGAN_model.py:147-160).  HBM-bound: reads p, g, m, v and writes p, m, v once — 28 bytes per parameter.
#include "esr_common.h"
#include <cstdlib>
#include <cstring>

namespace {
constexpr int ADAM_CHUNK = 4096;      // elements per workgroup
constexpr int ADAM_THREADS = 256;

struct AdamChunk { int32_t tensor; int32_t chunk; };

__global__ __launch_bounds__(ADAM_THREADS) void adam_multi_kernel(const esr_adam_tensor* __restrict__ tensors, const AdamChunk* __restrict__ chunks, float lr,
                                                                    float beta1, float beta2, float eps, float weight_decay, float bc1, float bc2_sqrt) {
    const AdamChunk c = chunks[blockIdx.x];
    const esr_adam_tensor t = tensors[c.tensor];
    const int64_t lo = (int64_t)c.chunk * ADAM_CHUNK;
    const int64_t hi = lo + ADAM_CHUNK < t.n ? lo + ADAM_CHUNK : t.n;
    const float step_size = lr / bc1;
    for (int64_t i = lo + threadIdx.x; i < hi; i += ADAM_THREADS) {
        float p = t.p[i], g = t.g[i], m = t.m[i], v = t.v[i];
        // the operation order of torch's implementation (torch/optim/adam.py, _multi_tensor_adam), each step rounded to fp32 as there
        if (weight_decay != 0.f) g = __fadd_rn(g, __fmul_rn(weight_decay, p));
        m = __fadd_rn(m, __fmul_rn(1.f - beta1, __fsub_rn(g, m)));                      // exp_avg.lerp_(grad, 1 - beta1)
        v = __fmaf_rn(__fmul_rn(g, g), 1.f - beta2, __fmul_rn(v, beta2));                // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), eps);
        p = __fadd_rn(p, __fmul_rn(-step_size, __fdiv_rn(m, denom)));                    // param.addcdiv_(exp_avg, denom, value=-step_size)
        t.p[i] = p; t.m[i] = m; t.v[i] = v;
    }
}
}  // namespace

static int64_t adam_nchunks(const esr_adam_tensor* t, int n) {
    int64_t c = 0;
    for (int i = 0; i < n; ++i) {
        if (!t[i].p || !t[i].g || !t[i].m || !t[i].v || t[i].n < 0) return ESR_E_ARG;
        c += (t[i].n + ADAM_CHUNK - 1) / ADAM_CHUNK;
    }
    return c;
}

extern "C" int64_t esr_adam_workspace_bytes(const esr_adam_tensor* tensors, int n) {
    if (!tensors || n <= 0) return ESR_E_ARG;
    const int64_t c = adam_nchunks(tensors, n);
    if (c < 0) return c;
    return (int64_t)n * sizeof(esr_adam_tensor) + c * (int64_t)sizeof(AdamChunk);
}

// the table as it lies in the workspace: [tensors | chunks]
static int64_t adam_fill(const esr_adam_tensor* tensors, int n, char* host) {
    memcpy(host, tensors, (size_t)n * sizeof(esr_adam_tensor));
    AdamChunk* ch = (AdamChunk*)(host + (size_t)n * sizeof(esr_adam_tensor));
    int64_t k = 0;
    for (int i = 0; i < n; ++i)
        for (int64_t j = 0; j * ADAM_CHUNK < tensors[i].n; ++j) ch[k++] = AdamChunk{i, (int32_t)j};
    return k;
}

extern "C" int64_t esr_adam_table(const esr_adam_tensor* tensors, int n, void* host_table, int64_t host_bytes) {
    const int64_t need = esr_adam_workspace_bytes(tensors, n);
    if (need < 0) return need;
    if (!host_table || host_bytes < need) return ESR_E_ARG;
    const int64_t c = adam_nchunks(tensors, n);
    if (c > 0x7fffffff) return ESR_E_UNSUPPORTED;
    return c == 0 ? 0 : adam_fill(tensors, n, (char*)host_table);
}

extern "C" int64_t esr_adam_upload(const esr_adam_tensor* tensors, int n, void* workspace, int64_t workspace_bytes, esr_stream_t stream) {
    const int64_t need = esr_adam_workspace_bytes(tensors, n);
    if (need < 0) return need;
    if (!workspace || workspace_bytes < need) return ESR_E_ARG;
    const int64_t c = adam_nchunks(tensors, n);
    if (c == 0) return 0;
    if (c > 0x7fffffff) return ESR_E_UNSUPPORTED;
    char* host = (char*)malloc((size_t)need);          // one host staging block
    if (!host) return ESR_E_ARG;
    adam_fill(tensors, n, host);
    hipError_t e = hipMemcpyAsync(workspace, host, (size_t)need, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);      // the staging block is freed on return
    free(host);
    return e == hipSuccess ? c : ESR_E_LAUNCH;
}

extern "C" int esr_adam_run(const void* workspace, int n, int64_t nchunks, float lr, float beta1, float beta2, float eps, float weight_decay, float bias_correction1,
                            float bias_correction2_sqrt, esr_stream_t stream) {
    if (!workspace || n <= 0 || nchunks < 0 || !(bias_correction1 > 0.f) || !(bias_correction2_sqrt > 0.f)) return ESR_E_ARG;
    if (nchunks == 0) return ESR_OK;
    const esr_adam_tensor* tensors = (const esr_adam_tensor*)workspace;
    const AdamChunk* chunks = (const AdamChunk*)((const char*)workspace + (size_t)n * sizeof(esr_adam_tensor));
    ESR_CLEAR_ERR();
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)nchunks), dim3(ADAM_THREADS), 0, (hipStream_t)stream, tensors, chunks, lr, beta1, beta2, eps, weight_decay,
                       bias_correction1, bias_correction2_sqrt);
    ESR_CHECK_LAUNCH();
    return ESR_OK;
}
