// libpcgc_reftable.so — the reference's CDF table, evaluated with the reference's own arithmetic, behind a C ABI.
//
// PCGCv2 derives the 8 x (L+1) table that decides every bit of `_F.bin` with torch on the CPU (entropy_model.py:82-101 `_logits_cumulative`,
// :112-130 `_likelihood`, :142-149 `_pmf_to_cdf`, :163-172 / :181-189 call sites) and hands it to torchac, which normalises it to 16 bits.
// torch's CPU kernels pick vectorised / scalar-tail / BLAS code by tensor shape and host, so "the same table" means: the same ATen
// operators, in the same order, on tensors of the same shape and layout.  This file is that operator sequence issued from C++ (no
// Python dispatch: ~60 tiny operators cost ~0.07 ms here against ~0.25 ms from Python, and the call sits on the critical path of every
// encode and decode).  pcgcv2_amd/entropy_model.py:reference_table is the same sequence in Python; tests pin both to golden tables
// generated from the reference.  Host only; links libtorch_cpu (the library the reference itself computes with).
#include <cmath>
#include <ATen/ATen.h>
#include <c10/core/InferenceMode.h>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

namespace {
struct ParamOps {
    std::vector<float> params;               // the bytes the entry was built from (the blobs below point into this copy)
    at::Tensor sp[4], bias[4], th[4];
};
std::mutex g_po_mu;
std::vector<std::shared_ptr<const ParamOps>> g_po;          // most recent last; a handful of models at most

std::shared_ptr<const ParamOps> param_ops(const float* params, int C) {
    const size_t n = (size_t)44 * C;
    {
        std::lock_guard<std::mutex> lk(g_po_mu);
        for (auto it = g_po.rbegin(); it != g_po.rend(); ++it)
            if ((*it)->params.size() == n && std::memcmp((*it)->params.data(), params, n * sizeof(float)) == 0) return *it;
    }
    const int F[5] = {1, 3, 3, 3, 1};
    const auto f32 = at::TensorOptions().dtype(at::kFloat);
    auto po = std::make_shared<ParamOps>();
    po->params.assign(params, params + n);
    const float* p = po->params.data();
    for (int i = 0; i < 4; ++i) {                                               // softplus(matrix_i)   entropy_model.py:94
        po->sp[i] = at::softplus(at::from_blob(const_cast<float*>(p), {C, F[i + 1], F[i]}, f32));
        p += (int64_t)C * F[i + 1] * F[i];
    }
    for (int i = 0; i < 4; ++i) { po->bias[i] = at::from_blob(const_cast<float*>(p), {C, F[i + 1], 1}, f32); p += (int64_t)C * F[i + 1]; }
    for (int i = 0; i < 4; ++i) {                                               // tanh(factor_i)       entropy_model.py:97
        po->th[i] = at::tanh(at::from_blob(const_cast<float*>(p), {C, F[i + 1], 1}, f32));
        p += (int64_t)C * F[i + 1];
    }
    std::lock_guard<std::mutex> lk(g_po_mu);
    if (g_po.size() >= 8) g_po.erase(g_po.begin());
    g_po.push_back(po);
    return po;
}
}  // namespace

// Drops the cached parameter-only operators (softplus / tanh of the 12 parameter tensors): with them gone the next table is evaluated
// from the raw parameters, operator for operator what the reference does on every call.  -> entries dropped
extern "C" int pcgc_reference_table_clear(void) {
    std::lock_guard<std::mutex> lk(g_po_mu);
    const int n = (int)g_po.size();
    g_po.clear();
    return n;
}

extern "C" int pcgc_reference_table(const float* params /*[host 44*C]: matrices 0..3 | biases 0..3 | factors 0..3*/, int C, float min_v,
                                    float max_v, uint16_t* table_u16 /*[host C, L+1]*/, float* cdf_f32 /*[host C, L+1] or NULL*/) {
    if (!params || !table_u16 || C < 1 || !(max_v >= min_v)) return -2;
    // the caller sizes the outputs as [C, (int)(max_v - min_v) + 2]; arange(min_v, max_v + 1) has that many entries only for integral
    // bounds (what compress() writes): a fractional bound from a damaged header would run past the buffer
    if (min_v != std::floor(min_v) || max_v != std::floor(max_v) || max_v - min_v > 65000.0f) return -2;
    try {
        c10::InferenceMode guard;
        const int F[5] = {1, 3, 3, 3, 1};
        const auto f32 = at::TensorOptions().dtype(at::kFloat);
        // softplus(matrix_i) (entropy_model.py:94) and tanh(factor_i) (:97) depend on the parameters only: evaluated once per parameter
        // set — the same operators on the same tensors, kept like any other re-laid-out weight — not once per table
        const std::shared_ptr<const ParamOps> po = param_ops(params, C);
        const at::Tensor* sp = po->sp;
        const at::Tensor* bias = po->bias;
        const at::Tensor* th = po->th;
        // symbols = arange(min_v, max_v + 1).reshape(-1, 1).repeat(1, C); inputs = symbols.permute(1, 0).contiguous().view(C, 1, -1):
        // a contiguous [C, 1, L] tensor whose every row is arange's values — built here as arange -> expand -> contiguous (same values,
        // same shape and strides; every operator below sees what it sees in the reference)
        at::Tensor sym = at::arange(at::Scalar((double)min_v), at::Scalar((double)max_v + 1), f32);
        const int64_t Lsym = sym.size(0);
        at::Tensor grid = sym.view({1, 1, Lsym}).expand({C, 1, Lsym}).contiguous();
        const std::vector<int64_t> shape = {C, Lsym};
        at::Tensor ends[2];
        const double half[2] = {-0.5, 0.5};                                      // lower = f(v - 0.5), upper = f(v + 0.5)
        for (int e = 0; e < 2; ++e) {
            at::Tensor z = grid + half[e];
            for (int i = 0; i < 4; ++i) {
                z = at::matmul(sp[i], z);
                z += bias[i];
                z += th[i] * at::tanh(z);
            }
            ends[e] = z;
        }
        at::Tensor sign = -at::sign(at::add(ends[0], ends[1]));
        at::Tensor lik = at::abs(at::sigmoid(sign * ends[1]) - at::sigmoid(sign * ends[0]));
        lik = lik.view(shape).permute({1, 0});                                   // [L, C] view, as _likelihood returns it
        at::Tensor pmf = at::clamp(lik, /*min=*/1e-9).permute({1, 0});           // [C, L]
        at::Tensor cdf = pmf.cumsum(-1);
        cdf = at::cat({at::zeros({pmf.size(0), 1}, f32), cdf}, -1).clamp(c10::nullopt, 1.0);
        // torchac 0.9.3 _convert_to_int_and_normalize: cdf.mul(2^16 - (Lp - 1)).round().to(int16).add_(arange(Lp, int16))
        const int64_t Lp = cdf.size(-1);
        at::Tensor top = at::scalar_tensor(2, f32).pow_(16) - (Lp - 1);
        at::Tensor q = cdf.mul(top).round().to(at::kShort);
        q.add_(at::arange(Lp, at::TensorOptions().dtype(at::kShort)));
        q = q.contiguous();
        std::memcpy(table_u16, q.data_ptr<int16_t>(), (size_t)C * Lp * sizeof(int16_t));
        if (cdf_f32) {
            at::Tensor cc = cdf.contiguous();
            std::memcpy(cdf_f32, cc.data_ptr<float>(), (size_t)C * Lp * sizeof(float));
        }
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}
