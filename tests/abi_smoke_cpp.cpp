// abi_smoke_cpp.cpp -- the C++ host mirror (playsnark_amd/host/playsnark.hpp) compiled and run: the toy circuit of
// r1cs.go:178-198 proved with playsnark::Groth16Prove against tests/golden/groth16_toy.json (flat "name hex" file
// from tests/test_abi.py), QAP::Quotient, Poly::BlindEval and the exceptions that stand for the reference's panics.
//   g++ -std=c++17 -Wall -I. tests/abi_smoke_cpp.cpp -Lplaysnark_amd -lplaysnark_hip -o abi_smoke_cpp
// Exit codes: 0 ok, 77 no gfx950 device, 1 failure.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include "playsnark_amd/host/playsnark.hpp"

using namespace playsnark;

static std::map<std::string, Bytes> load(const char* path) {
    std::map<std::string, Bytes> m;
    std::ifstream f(path);
    std::string name, hex;
    while (f >> name >> hex) {
        Bytes b(hex.size() / 2);
        for (size_t i = 0; i < b.size(); i++) b[i] = (uint8_t)std::stoi(hex.substr(2 * i, 2), nullptr, 16);
        m[name] = b;
    }
    return m;
}
#define REQUIRE(cond) do { if (!(cond)) { std::fprintf(stderr, "abi_smoke_cpp: line %d: %s failed\n", __LINE__, #cond); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    auto fx = load(argv[1]);
    try {
        Context ctx(0);
        QAP::Csr L{{0, 1, 2, 4, 6}, {1, 3, 1, 4, 0, 5}, {1, 1, 1, 1, 5, 1}};
        QAP::Csr R{{0, 1, 2, 3, 4}, {1, 1, 0, 0}, {1, 1, 1, 1}};
        QAP::Csr O{{0, 1, 2, 3, 4}, {3, 4, 5, 2}, {1, 1, 1, 1}};
        QAP qap(ctx, 6, 3, L, R, O);
        Points xi(ctx, PS_G1, fx["Xi"]), xi2(ctx, PS_G2, fx["Xi2"]), nio(ctx, PS_G1, fx["NioLP"]), xit(ctx, PS_G1, fx["XiT"]);
        ps_groth16_pk pk{};
        std::memcpy(pk.alpha, fx["Alpha"].data(), 96);
        std::memcpy(pk.beta, fx["Beta"].data(), 96);
        std::memcpy(pk.delta, fx["Delta"].data(), 96);
        std::memcpy(pk.beta2, fx["Beta2"].data(), 192);
        std::memcpy(pk.delta2, fx["Delta2"].data(), 192);
        pk.xi = xi.get(); pk.xi2 = xi2.get(); pk.nio_lp = nio.get(); pk.xi_t = xit.get();
        Poly sol = Poly::FromValues(ctx, {1, 3, 35, 9, 27, 30});
        Scalar r, s;
        std::memcpy(r.data(), fx["r"].data(), 32);
        std::memcpy(s.data(), fx["s"].data(), 32);
        Groth16Proof pf = Groth16Prove(ctx, pk, qap, sol, r, s);
        REQUIRE(!std::memcmp(pf.A.data(), fx["A"].data(), 96));
        REQUIRE(!std::memcmp(pf.B.data(), fx["B"].data(), 192));
        REQUIRE(!std::memcmp(pf.C.data(), fx["C"].data(), 96));
        Poly h = qap.Quotient(sol);
        REQUIRE(h.size() == 3 && !std::memcmp(h.Download()[0].data(), fx["h0"].data(), 32));
        REQUIRE(h.BlindEval(xit).size() == 96);
        bool threw = false;
        try { sol.BlindEval(xi); } catch (const LengthMismatch& e) { threw = std::strstr(e.what(), "mismatch of length between poly 6") != nullptr; }
        REQUIRE(threw);
        threw = false;
        try { qap.Quotient(Poly::FromValues(ctx, {1, 3, 35, 9, 27, 31})); } catch (const Apocalypse&) { threw = true; }
        REQUIRE(threw);
        Poly prod = Poly::FromValues(ctx, {1, 2}).Mul(Poly::FromValues(ctx, {3, 0, 1}));  // TestAlgebraPolyMul, algebra_test.go:76-104
        auto pc = prod.Download();
        const int want[4] = {3, 6, 1, 2};
        REQUIRE(pc.size() == 4);
        for (int i = 0; i < 4; i++) REQUIRE(pc[i][31] == want[i] && pc[i][0] == 0);
    } catch (const Error& e) {
        if (e.code == PS_ERR_NO_DEVICE) { std::printf("no gfx950 device: %s\n", e.what()); return 77; }
        std::fprintf(stderr, "abi_smoke_cpp: error %d: %s\n", e.code, e.what());
        return 1;
    }
    std::printf("abi_smoke_cpp ok\n");
    return 0;
}
