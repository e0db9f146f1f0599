// C++ host mirror of the reference's Go surface for the prover hot path, header-only over the C
// ABI (include/playsnark_hip.h).  The reference is compiled code (Go) whose toolchain is absent
// from the build image, so this is the host side "in the reference's shape": same names,
// argument meaning and error behaviour as algebra.go / qap.go / groth16.go / pinochio.go.
//
//   playsnark::Poly::BlindEval(points)          algebra.go:348-359
//   playsnark::Poly::Mul(p2)                    algebra.go:92-105
//   playsnark::QAP::Quotient(sol)               qap.go:151-162   (throws Apocalypse)
//   playsnark::Groth16Prove(tr, q, sol, r, s)   groth16.go:122-211
//   playsnark::PHGR13Prove(ek, qap, solution)   pinochio.go:207-254
//
// The reference panics; here the same conditions throw LengthMismatch (message of
// algebra.go:351) and Apocalypse ("apocalypse", qap.go:159).  No arithmetic in this file.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/playsnark_hip.h"

namespace playsnark {

struct LengthMismatch : std::runtime_error { using std::runtime_error::runtime_error; };
struct Apocalypse : std::runtime_error { Apocalypse() : std::runtime_error("apocalypse") {} };
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

inline void check(int rc) {
    if (rc == PS_OK) return;
    if (rc == PS_ERR_LENGTH) throw LengthMismatch(ps_last_error());
    if (rc == PS_ERR_NOT_DIVISIBLE) throw Apocalypse();
    throw Error(rc, ps_last_error());
}

using Bytes = std::vector<uint8_t>;
using Scalar = std::array<uint8_t, 32>;  // Element: 32-byte big-endian

class Context {
  public:
    explicit Context(int device = 0) { check(ps_ctx_create(device, &h_)); }
    ~Context() { ps_ctx_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    ps_ctx* get() const { return h_; }
    void sync() { check(ps_ctx_sync(h_)); }

  private:
    ps_ctx* h_ = nullptr;
};

// []G1 / []G2 resident on the device (Groth16Setup.Xi, PHGR13EvalKey.vs, ...)
class Points {
  public:
    Points(Context& c, int group, const Bytes& affine) : ctx_(&c) {
        const size_t wb = group == PS_G1 ? 96 : 192;
        check(ps_points_upload(c.get(), group, affine.data(), affine.size() / wb, PS_FMT_AFFINE, &h_));
    }
    Points(Points&& o) noexcept : ctx_(o.ctx_), h_(o.h_) { o.h_ = nullptr; }
    ~Points() { ps_points_free(h_); }
    Points(const Points&) = delete;
    Points& operator=(const Points&) = delete;
    const ps_points* get() const { return h_; }
    size_t size() const { return ps_points_len(h_); }
    int group() const { return ps_points_group(h_); }
    // This array read as {x^i P} (Xi, Xi2: nodes = 0; XiT, gsi: nodes = 1) -> {l_j(x) P} on the QAP's nodes, without the
    // secret point: the one-time conversion of a reference-made key (groth16.go:79-97) onto the prover's fast route.
    template <class Q>
    Points ToLagrange(const Q& qap, int nodes) const {
        ps_points* h = nullptr;
        check(ps_points_monomial_to_lagrange(ctx_->get(), qap.get(), h_, nodes, &h));
        return Points(*ctx_, h);
    }
    Bytes Download() const {
        Bytes out(size() * (group() == PS_G1 ? 96 : 192));
        if (!out.empty()) check(ps_points_download(ctx_->get(), h_, 0, size(), out.data()));
        return out;
    }

  private:
    Points(Context& c, ps_points* h) : ctx_(&c), h_(h) {}
    Context* ctx_;
    ps_points* h_ = nullptr;
};

// type Poly []Element (algebra.go:89); also Vector via FromValues (Value.ToFieldElement, curve.go:17-19)
class Poly {
  public:
    Poly(Context& c, const std::vector<Scalar>& coeffs) : ctx_(&c) {
        check(ps_scalars_upload(c.get(), coeffs.empty() ? nullptr : coeffs[0].data(), coeffs.size(), &h_));
    }
    static Poly FromValues(Context& c, const std::vector<int64_t>& v) {
        ps_scalars* h = nullptr;
        check(ps_scalars_upload_i64(c.get(), v.data(), v.size(), &h));
        return Poly(c, h);
    }
    Poly(Poly&& o) noexcept : ctx_(o.ctx_), h_(o.h_) { o.h_ = nullptr; }
    ~Poly() { ps_scalars_free(h_); }
    Poly(const Poly&) = delete;
    Poly& operator=(const Poly&) = delete;
    const ps_scalars* get() const { return h_; }
    size_t size() const { return ps_scalars_len(h_); }

    // func (p Poly) BlindEval(zero Commit, blindedPoint []Commit) Commit
    Bytes BlindEval(const Points& blindedPoint) const {
        Bytes out(blindedPoint.group() == PS_G1 ? 96 : 192);
        check(ps_msm(ctx_->get(), blindedPoint.get(), h_, out.data()));
        return out;
    }
    // func (p Poly) Mul(p2 Poly) Poly
    Poly Mul(const Poly& p2) const {
        ps_scalars* h = nullptr;
        check(ps_poly_mul(ctx_->get(), h_, p2.h_, &h));
        return Poly(*ctx_, h);
    }
    std::vector<Scalar> Download() const {
        std::vector<Scalar> out(size());
        if (!out.empty()) check(ps_scalars_download(ctx_->get(), h_, 0, out.size(), out[0].data()));
        return out;
    }

  private:
    friend class QAP;
    Poly(Context& c, ps_scalars* h) : ctx_(&c), h_(h) {}
    Context* ctx_;
    ps_scalars* h_ = nullptr;
};

// type QAP (qap.go:10-27) in sparse evaluation form on the reference's domain {1..n}
class QAP {
  public:
    struct Csr {
        std::vector<uint32_t> row_ptr, col;
        std::vector<int64_t> val;
    };
    QAP(Context& c, size_t nbVars, size_t nbIO, const Csr& left, const Csr& right, const Csr& out) : ctx_(&c) {
        ps_csr l{left.row_ptr.data(), left.col.data(), left.val.data()};
        ps_csr r{right.row_ptr.data(), right.col.data(), right.val.data()};
        ps_csr o{out.row_ptr.data(), out.col.data(), out.val.data()};
        check(ps_qap_create(c.get(), left.row_ptr.size() - 1, nbVars, nbIO, &l, &r, &o, &h_));
    }
    ~QAP() { ps_qap_free(h_); }
    QAP(const QAP&) = delete;
    QAP& operator=(const QAP&) = delete;
    const ps_qap* get() const { return h_; }
    // func (q QAP) Quotient(sol Vector) Poly -- throws Apocalypse on a non-zero remainder
    Poly Quotient(const Poly& sol) const {
        ps_scalars* h = nullptr;
        check(ps_qap_quotient(ctx_->get(), h_, sol.get(), nullptr, nullptr, nullptr, &h));
        return Poly(*ctx_, h);
    }

  private:
    Context* ctx_;
    ps_qap* h_ = nullptr;
};

struct Groth16Proof {  // groth16.go:106-118
    Scalar R, S;
    std::array<uint8_t, 96> A;
    std::array<uint8_t, 192> B;
    std::array<uint8_t, 96> C;
};

// func Groth16Prove(tr Groth16Setup, q QAP, sol Vector) Groth16Proof; r, s drawn by the caller
inline Groth16Proof Groth16Prove(Context& c, const ps_groth16_pk& tr, const QAP& q, const Poly& sol, const Scalar& r,
                                 const Scalar& s) {
    Groth16Proof p{r, s, {}, {}, {}};
    check(ps_groth16_prove(c.get(), &tr, q.get(), sol.get(), r.data(), s.data(), p.A.data(), p.B.data(), p.C.data()));
    return p;
}

// One rank's share of Groth16Prove when the sums are split over `world` GPUs; the ranks' A, B, C add up
// (ps_points_sum after an all_gather) to the proof
inline Groth16Proof Groth16ProveShard(Context& c, const ps_groth16_pk& tr, const QAP& q, const Poly& sol, const Scalar& r,
                                      const Scalar& s, int rank, int world) {
    Groth16Proof p{r, s, {}, {}, {}};
    check(ps_groth16_prove_shard(c.get(), &tr, q.get(), sol.get(), r.data(), s.data(), rank, world, p.A.data(), p.B.data(),
                                 p.C.data()));
    return p;
}

// func PHGR13Prove(ek PHGR13EvalKey, qap QAP, solution Vector) PHGR13Proof
inline ps_phgr13_proof PHGR13Prove(Context& c, const ps_phgr13_ek& ek, const QAP& qap, const Poly& solution) {
    ps_phgr13_proof out;
    check(ps_phgr13_prove(c.get(), &ek, qap.get(), solution.get(), &out));
    return out;
}

// func Groth16Verify(tr Groth16Setup, q QAP, p Groth16Proof, io Vector) bool (groth16.go:214)
inline bool Groth16Verify(Context& c, const ps_groth16_vk& vk, const Groth16Proof& p, const Poly& io) {
    int ok = 0;
    check(ps_groth16_verify(c.get(), &vk, io.get(), p.A.data(), p.B.data(), p.C.data(), &ok));
    return ok != 0;
}
// func PHGR13Verify(vk PHGR13VerifKey, qap QAP, p PHGR13Proof, io Vector) bool (pinochio.go:281)
inline bool PHGR13Verify(Context& c, const ps_phgr13_vk& vk, const ps_phgr13_proof& p, const Poly& io) {
    int ok = 0;
    check(ps_phgr13_verify(c.get(), &vk, io.get(), &p, &ok));
    return ok != 0;
}
// func NewGroth16TrustedSetup(qap QAP) Groth16Setup (groth16.go:64), toxic waste drawn by the caller
inline ps_groth16_crs NewGroth16TrustedSetup(Context& c, const QAP& q, const ps_groth16_toxic& tw) {
    ps_groth16_crs out;
    check(ps_groth16_setup(c.get(), q.get(), &tw, &out));
    return out;
}

// func NewPHGR13TrustedSetup(qap QAP) PHGR13Setup (pinochio.go:93), toxic waste drawn by the caller;
// release the arrays with ps_phgr13_crs_free
inline ps_phgr13_crs NewPHGR13TrustedSetup(Context& c, const QAP& q, const ps_phgr13_toxic& tw) {
    ps_phgr13_crs out;
    check(ps_phgr13_setup(c.get(), q.get(), &tw, &out));
    return out;
}
// computeSolCommit for several evaluation-key arrays at once (pinochio.go:222-241): one digit sort
inline std::vector<Bytes> SolCommits(Context& c, const std::vector<const Points*>& arrays, const Poly& sol) {
    std::vector<const ps_points*> pts;
    std::vector<Bytes> out;
    std::vector<uint8_t*> dst;
    for (auto* a : arrays) {
        pts.push_back(a->get());
        out.emplace_back(a->group() == PS_G1 ? 96 : 192);
    }
    for (auto& o : out) dst.push_back(o.data());
    check(ps_msm_multi(c.get(), pts.data(), pts.size(), sol.get(), dst.data()));
    return out;
}

}  // namespace playsnark
