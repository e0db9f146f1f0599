/* abi_smoke.c -- the C ABI used the way a cgo (or any other FFI) caller uses it: plain C99, no Python,
 * no ctypes.  Proves the reference's toy circuit x^3 + x + 5 = 35 (r1cs.go:178-198, witness r1cs.go:67-76)
 * with Groth16Prove (groth16.go:122-211) against the committed golden fixture tests/golden/groth16_toy.json
 * (handed over by tests/test_abi.py as a flat "name hex" text file), checks the proof with ps_groth16_verify,
 * and exercises the error codes that stand for the reference's panics.
 *
 *   gcc -std=c99 -Wall -Iinclude tests/abi_smoke.c -Lplaysnark_amd -lplaysnark_hip -o abi_smoke
 *   ./abi_smoke fixture.txt
 * Exit codes: 0 = all checks passed, 77 = no gfx950 device (the library has no CPU fallback), 1 = failure.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "playsnark_hip.h"

#define MAXF 32
static struct { char name[32]; uint8_t* data; size_t len; } fx[MAXF];
static int nfx = 0;

static int hexval(int c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : -1; }

static int load_fixture(const char* path) {
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    static char line[1 << 16];
    while (fgets(line, sizeof line, f) && nfx < MAXF) {
        char* sp = strchr(line, ' ');
        if (!sp) continue;
        *sp++ = 0;
        size_t hl = strcspn(sp, "\r\n");
        strncpy(fx[nfx].name, line, sizeof fx[nfx].name - 1);
        fx[nfx].len = hl / 2;
        fx[nfx].data = (uint8_t*)malloc(hl / 2 + 1);
        for (size_t i = 0; i < hl / 2; i++) fx[nfx].data[i] = (uint8_t)(hexval(sp[2 * i]) << 4 | hexval(sp[2 * i + 1]));
        nfx++;
    }
    fclose(f);
    return 0;
}
static const uint8_t* get(const char* name, size_t want_len) {
    for (int i = 0; i < nfx; i++)
        if (!strcmp(fx[i].name, name)) {
            if (fx[i].len != want_len) { fprintf(stderr, "fixture %s: %zu bytes, expected %zu\n", name, fx[i].len, want_len); exit(1); }
            return fx[i].data;
        }
    fprintf(stderr, "fixture %s missing\n", name);
    exit(1);
}

#define CHECK(cond)                                                                              \
    do {                                                                                         \
        if (!(cond)) {                                                                           \
            fprintf(stderr, "abi_smoke: %s:%d: %s failed (last error: %s)\n", __FILE__, __LINE__, #cond, ps_last_error()); \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2 || load_fixture(argv[1])) { fprintf(stderr, "usage: abi_smoke fixture.txt\n"); return 1; }
    printf("%s, %d device(s)\n", ps_version(), ps_device_count());
    CHECK(ps_abi_version() == PS_ABI_VERSION); /* the library was built from THIS header revision */
    CHECK(sizeof(ps_msm_info) == 32 && sizeof(ps_groth16_pk) == 3 * 96 + 2 * 192 + 7 * sizeof(void*));
    ps_ctx* ctx = NULL;
    int rc = ps_ctx_create(0, &ctx);
    if (rc == PS_ERR_NO_DEVICE) { printf("no gfx950 device: %s\n", ps_last_error()); return 77; }
    CHECK(rc == PS_OK);

    /* the toy R1CS, rows = gates, columns = [const, x, out, u, v, w] (r1cs.go:178-198) */
    const uint32_t l_ptr[5] = {0, 1, 2, 4, 6}, l_col[6] = {1, 3, 1, 4, 0, 5};
    const int64_t l_val[6] = {1, 1, 1, 1, 5, 1};
    const uint32_t r_ptr[5] = {0, 1, 2, 3, 4}, r_col[4] = {1, 1, 0, 0};
    const int64_t r_val[4] = {1, 1, 1, 1};
    const uint32_t o_ptr[5] = {0, 1, 2, 3, 4}, o_col[4] = {3, 4, 5, 2};
    const int64_t o_val[4] = {1, 1, 1, 1};
    const ps_csr L = {l_ptr, l_col, l_val}, R = {r_ptr, r_col, r_val}, O = {o_ptr, o_col, o_val};
    ps_qap* qap = NULL;
    CHECK(ps_qap_create(ctx, 4, 6, 3, &L, &R, &O, &qap) == PS_OK);

    /* CRS of the fixture: n = 4 gates, nbVars - nbIO = 3 */
    ps_points *xi = NULL, *xi2 = NULL, *nio = NULL, *xit = NULL, *iolp = NULL;
    CHECK(ps_points_upload(ctx, PS_G1, get("Xi", 4 * 96), 4, PS_FMT_AFFINE, &xi) == PS_OK);
    CHECK(ps_points_upload(ctx, PS_G2, get("Xi2", 4 * 192), 4, PS_FMT_AFFINE, &xi2) == PS_OK);
    CHECK(ps_points_upload(ctx, PS_G1, get("NioLP", 3 * 96), 3, PS_FMT_AFFINE, &nio) == PS_OK);
    CHECK(ps_points_upload(ctx, PS_G1, get("XiT", 3 * 96), 3, PS_FMT_AFFINE, &xit) == PS_OK);
    CHECK(ps_points_upload(ctx, PS_G1, get("IoLP", 3 * 96), 3, PS_FMT_AFFINE, &iolp) == PS_OK);
    CHECK(ps_points_len(xi) == 4 && ps_points_group(xi2) == PS_G2);
    int in_subgroup = 0;
    CHECK(ps_points_check_subgroup(ctx, xi2, &in_subgroup) == PS_OK && in_subgroup == 1);

    ps_groth16_pk pk;
    memset(&pk, 0, sizeof pk);
    memcpy(pk.alpha, get("Alpha", 96), 96);
    memcpy(pk.beta, get("Beta", 96), 96);
    memcpy(pk.delta, get("Delta", 96), 96);
    memcpy(pk.beta2, get("Beta2", 192), 192);
    memcpy(pk.delta2, get("Delta2", 192), 192);
    pk.xi = xi; pk.xi2 = xi2; pk.nio_lp = nio; pk.xi_t = xit;

    const int64_t witness[6] = {1, 3, 35, 9, 27, 30}; /* createWitness, r1cs.go:67-76 */
    ps_scalars* sol = NULL;
    CHECK(ps_scalars_upload_i64(ctx, witness, 6, &sol) == PS_OK);

    uint8_t A[96], B[192], C[96], comp[96];
    CHECK(ps_groth16_prove(ctx, &pk, qap, sol, get("r", 32), get("s", 32), A, B, C) == PS_OK);
    CHECK(!memcmp(A, get("A", 96), 96));
    CHECK(!memcmp(B, get("B", 192), 192));
    CHECK(!memcmp(C, get("C", 96), 96));
    /* what the Go side feeds to UnmarshalBinary */
    CHECK(ps_point_convert(PS_G1, PS_FMT_AFFINE, PS_FMT_COMPRESSED, A, comp) == PS_OK && !memcmp(comp, get("A_compressed", 48), 48));
    CHECK(ps_point_convert(PS_G2, PS_FMT_AFFINE, PS_FMT_COMPRESSED, B, comp) == PS_OK && !memcmp(comp, get("B_compressed", 96), 96));
    CHECK(ps_point_convert(PS_G1, PS_FMT_AFFINE, PS_FMT_COMPRESSED, C, comp) == PS_OK && !memcmp(comp, get("C_compressed", 48), 48));
    uint8_t back[192];
    CHECK(ps_point_convert(PS_G2, PS_FMT_COMPRESSED, PS_FMT_AFFINE, get("B_compressed", 96), back) == PS_OK && !memcmp(back, B, 192));

    /* The reference-made key (monomial arrays only) onto the route without interpolation and division, WITHOUT its toxic waste
     * (groth16.go:13-14): Xi, Xi2 over the nodes 1..n, XiT over n+1..2n-1, converted over the group elements alone -- the shim's
     * (*HipGroth16).ToLagrange().  The proof bytes do not change. */
    {
        ps_points *lxi = NULL, *lxi2 = NULL, *lxit = NULL, *bad_len = NULL;
        CHECK(ps_points_monomial_to_lagrange(ctx, qap, xi, 0, &lxi) == PS_OK && ps_points_len(lxi) == 4);
        CHECK(ps_points_monomial_to_lagrange(ctx, qap, xi2, 0, &lxi2) == PS_OK && ps_points_group(lxi2) == PS_G2);
        CHECK(ps_points_monomial_to_lagrange(ctx, qap, xit, 1, &lxit) == PS_OK && ps_points_len(lxit) == 3);
        CHECK(ps_points_monomial_to_lagrange(ctx, qap, xit, 0, &bad_len) == PS_ERR_LENGTH && bad_len == NULL);
        ps_groth16_pk fast = pk;
        fast.lxi = lxi; fast.lxi2 = lxi2; fast.lxi_t = lxit;
        uint8_t A2[96], B2[192], C2[96];
        CHECK(ps_groth16_prove(ctx, &fast, qap, sol, get("r", 32), get("s", 32), A2, B2, C2) == PS_OK);
        CHECK(!memcmp(A2, A, 96) && !memcmp(B2, B, 192) && !memcmp(C2, C, 96));
        ps_points_free(lxi); ps_points_free(lxi2); ps_points_free(lxit);
    }

    /* Groth16Verify (groth16.go:214-233): Gamma = gamma * G2 from the fixture's toxic waste */
    ps_scalars* gam = NULL;
    ps_points* gam_pt = NULL;
    CHECK(ps_scalars_upload(ctx, get("gamma", 32), 1, &gam) == PS_OK);
    CHECK(ps_points_from_scalars(ctx, PS_G2, gam, &gam_pt) == PS_OK);
    ps_groth16_vk vk;
    memset(&vk, 0, sizeof vk);
    memcpy(vk.alpha, pk.alpha, 96);
    memcpy(vk.beta2, pk.beta2, 192);
    memcpy(vk.delta2, pk.delta2, 192);
    CHECK(ps_points_download(ctx, gam_pt, 0, 1, vk.gamma) == PS_OK);
    vk.io_lp = iolp;
    ps_scalars* io = NULL;
    CHECK(ps_scalars_slice(sol, 0, 3, &io) == PS_OK); /* sol[:diff], groth16_test.go:29 */
    int ok = 0;
    CHECK(ps_groth16_verify(ctx, &vk, io, A, B, C, &ok) == PS_OK && ok == 1);
    CHECK(ps_groth16_verify(ctx, &vk, io, A, B, get("Alpha", 96), &ok) == PS_OK && ok == 0); /* a wrong C */

    /* Poly.BlindEval's panic (algebra.go:350-352) and QAP.Quotient's (qap.go:158-160) as error codes */
    uint8_t out[96];
    CHECK(ps_msm(ctx, xi, io, out) == PS_ERR_LENGTH);
    CHECK(strstr(ps_last_error(), "mismatch of length between poly 3 and blinded eval points 4") != NULL);
    const int64_t bad_witness[6] = {1, 3, 35, 9, 27, 31};
    ps_scalars* bad = NULL;
    CHECK(ps_scalars_upload_i64(ctx, bad_witness, 6, &bad) == PS_OK);
    ps_scalars* h = NULL;
    CHECK(ps_qap_quotient(ctx, qap, bad, NULL, NULL, NULL, &h) == PS_ERR_NOT_DIVISIBLE);
    CHECK(ps_qap_quotient(ctx, qap, sol, NULL, NULL, NULL, &h) == PS_OK && ps_scalars_len(h) == 3);
    uint8_t h0[32];
    CHECK(ps_scalars_download(ctx, h, 0, 1, h0) == PS_OK && !memcmp(h0, get("h0", 32), 32)); /* h0 = -11/3 */

    /* a sum through the asynchronous queue: h . XiT (groth16.go:185) */
    CHECK(ps_msm_launch(ctx, xit, h) == PS_OK && ps_msm_finish(ctx, out) == PS_OK);
    uint8_t out2[96];
    CHECK(ps_msm(ctx, xit, h, out2) == PS_OK && !memcmp(out, out2, 96));

    /* (*QAP).IsValid, qap.go:107-148 */
    int valid = -1;
    CHECK(ps_qap_is_valid(ctx, qap, sol, &valid) == PS_OK && valid == 1);
    CHECK(ps_qap_is_valid(ctx, qap, bad, &valid) == PS_OK && valid == 0);
    ps_msm_info info;
    CHECK(ps_msm_last_info(ctx, &info) == PS_OK && info.windows > 0 && info.window_table == 0);

    /* ---- the in-process multi-device entries, as a cgo caller uses them (shim/playsnark_hip.go: BlindEvalHIPMulti,
     * Groth16ProveHIPMulti), here with two contexts on device 0.  Device d holds ITS index range of every array. ---- */
    {
        ps_ctx* ctx2 = NULL;
        CHECK(ps_ctx_create(0, &ctx2) == PS_OK);
        ps_ctx* ctxs[2];
        ctxs[0] = ctx; ctxs[1] = ctx2;
        /* h . XiT over two shards: XiT[0..2) on context 0, XiT[2..3) on context 1 (sizes differ by at most one) */
        ps_points *xit0 = NULL, *xit1 = NULL;
        ps_scalars *h0s = NULL, *h1s = NULL;
        uint8_t hraw[3 * 32];
        CHECK(ps_scalars_download(ctx, h, 0, 3, hraw) == PS_OK);
        CHECK(ps_points_upload(ctx, PS_G1, get("XiT", 3 * 96), 2, PS_FMT_AFFINE, &xit0) == PS_OK);
        CHECK(ps_points_upload(ctx2, PS_G1, get("XiT", 3 * 96) + 2 * 96, 1, PS_FMT_AFFINE, &xit1) == PS_OK);
        CHECK(ps_scalars_upload(ctx, hraw, 2, &h0s) == PS_OK && ps_scalars_upload(ctx2, hraw + 64, 1, &h1s) == PS_OK);
        const ps_points* mp[2];
        const ps_scalars* ms[2];
        mp[0] = xit0; mp[1] = xit1; ms[0] = h0s; ms[1] = h1s;
        uint8_t out3[96];
        CHECK(ps_msm_multi_device(ctxs, mp, ms, 2, out3) == PS_OK && !memcmp(out3, out, 96));

        /* Groth16Prove over rank-local keys: Xi, Xi2 (4 points) 2 + 2; NioLP, XiT (3 points) 2 + 1 */
        ps_groth16_device dev[2];
        memset(dev, 0, sizeof dev); /* the header requires zero-initialised structs */
        ps_qap* qap2 = NULL;
        ps_scalars* sol2 = NULL;
        CHECK(ps_qap_create(ctx2, 4, 6, 3, &L, &R, &O, &qap2) == PS_OK);
        CHECK(ps_scalars_upload_i64(ctx2, witness, 6, &sol2) == PS_OK);
        ps_points* part[2][4];
        for (int d = 0; d < 2; d++) {
            ps_ctx* cd = ctxs[d];
            const size_t f4 = d ? 2 : 0, n4 = 2, f3 = d ? 2 : 0, n3 = d ? 1 : 2;
            CHECK(ps_points_upload(cd, PS_G1, get("Xi", 4 * 96) + f4 * 96, n4, PS_FMT_AFFINE, &part[d][0]) == PS_OK);
            CHECK(ps_points_upload(cd, PS_G2, get("Xi2", 4 * 192) + f4 * 192, n4, PS_FMT_AFFINE, &part[d][1]) == PS_OK);
            CHECK(ps_points_upload(cd, PS_G1, get("NioLP", 3 * 96) + f3 * 96, n3, PS_FMT_AFFINE, &part[d][2]) == PS_OK);
            CHECK(ps_points_upload(cd, PS_G1, get("XiT", 3 * 96) + f3 * 96, n3, PS_FMT_AFFINE, &part[d][3]) == PS_OK);
            dev[d].ctx = cd;
            dev[d].qap = d ? qap2 : qap;
            dev[d].sol = d ? sol2 : sol;
            dev[d].pk.xi = part[d][0]; dev[d].pk.xi2 = part[d][1]; dev[d].pk.nio_lp = part[d][2]; dev[d].pk.xi_t = part[d][3];
        }
        memcpy(dev[0].pk.alpha, pk.alpha, 96); memcpy(dev[0].pk.beta, pk.beta, 96); memcpy(dev[0].pk.delta, pk.delta, 96);
        memcpy(dev[0].pk.beta2, pk.beta2, 192); memcpy(dev[0].pk.delta2, pk.delta2, 192);
        uint8_t A2[96], B2[192], C2[96];
        CHECK(ps_groth16_prove_multi(dev, 2, get("r", 32), get("s", 32), A2, B2, C2) == PS_OK);
        CHECK(!memcmp(A2, get("A", 96), 96) && !memcmp(B2, get("B", 192), 192) && !memcmp(C2, get("C", 96), 96));
        /* a key that is not cut at the index ranges is refused, as BlindEval's length panic */
        dev[1].pk.xi_t = part[1][0]; /* 2 points where device 1 must hold 1 */
        CHECK(ps_groth16_prove_multi(dev, 2, get("r", 32), get("s", 32), A2, B2, C2) == PS_ERR_LENGTH);
        for (int d = 0; d < 2; d++)
            for (int k = 0; k < 4; k++) ps_points_free(part[d][k]);
        ps_points_free(xit0); ps_points_free(xit1); ps_scalars_free(h0s); ps_scalars_free(h1s); ps_scalars_free(sol2);
        ps_qap_free(qap2);
        ps_ctx_destroy(ctx2);
    }

    ps_scalars_free(h); ps_scalars_free(bad); ps_scalars_free(io); ps_scalars_free(gam); ps_scalars_free(sol);
    ps_points_free(gam_pt); ps_points_free(xi); ps_points_free(xi2); ps_points_free(nio); ps_points_free(xit); ps_points_free(iolp);
    ps_qap_free(qap);
    ps_ctx_destroy(ctx);
    printf("abi_smoke ok: toy Groth16 proof equals the golden fixture (one context, and two contexts over rank-local keys), verifies, and the error codes map the reference's panics\n");
    return 0;
}
