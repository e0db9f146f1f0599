// +build hip

// playsnark_hip.go -- cgo binding of libplaysnark_hip.so (include/playsnark_hip.h) for the
// nikkolasg/playsnark package.  Drop this file into the package directory and build with
// `go build -tags hip` (CGO_CFLAGS / CGO_LDFLAGS pointing at include/ and playsnark_amd/).
//
// It lives INSIDE package playsnark because the reference's key types have unexported fields
// (PHGR13EvalKey.vs .., QAP.nbVars ..).  The Go side only ever uses kyber's MarshalBinary /
// UnmarshalBinary (the reference's one serialisation site is pinochio.go:258-272): points cross the
// boundary in the ZCash compressed form those methods speak (PS_FMT_COMPRESSED on upload, batch
// decompression on the GPU; ps_point_convert on the few proof elements that come back).
//
// The image this library is built in has no Go toolchain, so this file is source only (SURVEY.md
// 8 row f4); every C entry point it calls is exercised by tests/abi_smoke.c (plain C) and by the
// ctypes binding the test-suite runs on.
package playsnark

/*
#cgo CFLAGS: -I${SRCDIR}/playsnark-hip/include
#cgo LDFLAGS: -L${SRCDIR}/playsnark-hip/playsnark_amd -lplaysnark_hip -Wl,-rpath,${SRCDIR}/playsnark-hip/playsnark_amd
#include <stdlib.h>
#include <string.h>
#include "playsnark_hip.h"
*/
import "C"

import (
	"fmt"
	"unsafe"

	"github.com/drand/kyber/util/random"
)

// ---------------------------------------------------------------------------------------
// context and error mapping
// ---------------------------------------------------------------------------------------

var hipCtx *C.ps_ctx

func init() {
	if rc := C.ps_ctx_create(0, &hipCtx); rc != C.PS_OK {
		// no CPU fallback behind this build tag: fail as loudly as the reference's own panics
		panic("playsnark_hip: " + C.GoString(C.ps_last_error()))
	}
}

// check maps the C error codes onto the reference's panics.
func check(rc C.int) {
	switch rc {
	case C.PS_OK:
	case C.PS_ERR_NOT_DIVISIBLE:
		panic("apocalypse") // qap.go:159, pinochio.go:215
	default:
		// PS_ERR_LENGTH carries the message of algebra.go:351, PS_ERR_ARG that of sanityCheck (qap.go:177-189)
		panic(C.GoString(C.ps_last_error()))
	}
}

func u8(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

// ---------------------------------------------------------------------------------------
// byte forms: only MarshalBinary / UnmarshalBinary on the Go side
// ---------------------------------------------------------------------------------------

const (
	g1Wire = 96 // PS_FMT_AFFINE sizes; MarshalBinary emits half of these (compressed)
	g2Wire = 192
)

func wireLen(group C.int) int {
	if group == C.PS_G1 {
		return g1Wire
	}
	return g2Wire
}

func marshalPoints(pts []Commit) []byte {
	out := make([]byte, 0, len(pts)*g2Wire/2)
	for _, p := range pts {
		b, err := p.MarshalBinary() // 48 B (G1) / 96 B (G2), ZCash compressed
		if err != nil {
			panic(err)
		}
		out = append(out, b...)
	}
	return out
}

func marshalScalars(p Poly) []byte {
	out := make([]byte, 0, 32*len(p))
	for _, e := range p {
		b, err := e.MarshalBinary() // 32-byte big-endian (kyber mod.Int)
		if err != nil {
			panic(err)
		}
		out = append(out, b...)
	}
	return out
}

// affineOf gives the uncompressed bytes the fixed-point fields of the C structs take.
func affineOf(group C.int, p Commit) []byte {
	in, err := p.MarshalBinary()
	if err != nil {
		panic(err)
	}
	out := make([]byte, wireLen(group))
	check(C.ps_point_convert(group, C.PS_FMT_COMPRESSED, C.PS_FMT_AFFINE, u8(in), u8(out)))
	return out
}

// pointFrom turns an affine result of the library back into a kyber point of the group of `like`.
func pointFrom(group C.int, affine []byte, like Commit) Commit {
	comp := make([]byte, wireLen(group)/2)
	check(C.ps_point_convert(group, C.PS_FMT_AFFINE, C.PS_FMT_COMPRESSED, u8(affine), u8(comp)))
	p := like.Clone()
	if err := p.UnmarshalBinary(comp); err != nil {
		panic(err)
	}
	return p
}

func copyTo(dst unsafe.Pointer, src []byte) {
	C.memcpy(dst, unsafe.Pointer(&src[0]), C.size_t(len(src)))
}

// uploadPoints keeps a CRS slice resident in HBM (once per setup, reused by every proof).  The byte
// slice is not retained by the library past the call (cgo pointer rules).
func uploadPoints(group C.int, pts []Commit) *C.ps_points {
	raw := marshalPoints(pts)
	var h *C.ps_points
	check(C.ps_points_upload(hipCtx, group, u8(raw), C.size_t(len(pts)), C.PS_FMT_COMPRESSED, &h))
	return h
}

// uploadSolution: Vector = []Value = []int (algebra.go:13); Value.ToFieldElement is SetInt64 (curve.go:17-19).
func uploadSolution(sol Vector) *C.ps_scalars {
	vals := make([]C.int64_t, len(sol))
	for i, v := range sol {
		vals[i] = C.int64_t(v)
	}
	var h *C.ps_scalars
	var p *C.int64_t
	if len(vals) > 0 {
		p = &vals[0]
	}
	check(C.ps_scalars_upload_i64(hipCtx, p, C.size_t(len(vals)), &h))
	return h
}

func downloadPoly(h *C.ps_scalars) Poly {
	n := int(C.ps_scalars_len(h))
	raw := make([]byte, 32*n)
	check(C.ps_scalars_download(hipCtx, h, 0, C.size_t(n), u8(raw)))
	out := make(Poly, n)
	for i := range out {
		e := NewElement()
		if err := e.UnmarshalBinary(raw[32*i : 32*i+32]); err != nil {
			panic(err)
		}
		out[i] = e
	}
	return out
}

// ---------------------------------------------------------------------------------------
// Poly.BlindEval (algebra.go:348-359) and Poly.Mul (algebra.go:92-105)
// ---------------------------------------------------------------------------------------

func groupOf(zero Commit) C.int {
	// the group is the DYNAMIC type of the points (declared Go types lie: Xi2 []G1 at groth16.go:60)
	if len(mustMarshal(zero)) == g1Wire/2 {
		return C.PS_G1
	}
	return C.PS_G2
}

func mustMarshal(p Commit) []byte {
	b, err := p.MarshalBinary()
	if err != nil {
		panic(err)
	}
	return b
}

// BlindEvalHIP replaces `func (p Poly) BlindEval(zero Commit, blindedPoint []Commit) Commit` for a
// CRS slice that was uploaded once with uploadPoints.
func (p Poly) BlindEvalHIP(zero Commit, crs *C.ps_points) Commit {
	if int(C.ps_points_len(crs)) != len(p) { // the reference's own panic and message, algebra.go:350-352
		panic(fmt.Sprintf("mismatch of length between poly %d and blinded eval points %d", len(p), int(C.ps_points_len(crs))))
	}
	group := C.ps_points_group(crs)
	sc := marshalScalars(p)
	out := make([]byte, wireLen(group))
	check(C.ps_msm_be32(hipCtx, crs, u8(sc), C.size_t(len(p)), u8(out)))
	return pointFrom(group, out, zero)
}

// MulHIP replaces `func (p Poly) Mul(p2 Poly) Poly`.
func (p Poly) MulHIP(p2 Poly) Poly {
	var a, b, prod *C.ps_scalars
	ra, rb := marshalScalars(p), marshalScalars(p2)
	check(C.ps_scalars_upload(hipCtx, u8(ra), C.size_t(len(p)), &a))
	defer C.ps_scalars_free(a)
	check(C.ps_scalars_upload(hipCtx, u8(rb), C.size_t(len(p2)), &b))
	defer C.ps_scalars_free(b)
	check(C.ps_poly_mul(hipCtx, a, b, &prod))
	defer C.ps_scalars_free(prod)
	return downloadPoly(prod)
}

// ---------------------------------------------------------------------------------------
// QAP: the R1CS matrices (r1cs.go:78-101, rows = gates, columns = variables) in CSR.  The arrays a
// ps_csr points at must not be Go memory holding Go pointers, so they are staged in C memory.
// ---------------------------------------------------------------------------------------

type cCsr struct {
	csr         C.ps_csr
	rp, col, va unsafe.Pointer
}

func newCsr(m Matrix) *cCsr {
	nnz := 0
	for _, row := range m {
		for _, v := range row {
			if v != 0 {
				nnz++
			}
		}
	}
	c := &cCsr{
		rp:  C.malloc(C.size_t(4 * (len(m) + 1))),
		col: C.malloc(C.size_t(4*nnz + 4)),
		va:  C.malloc(C.size_t(8*nnz + 8)),
	}
	rp := (*[1 << 30]C.uint32_t)(c.rp)[: len(m)+1 : len(m)+1]
	col := (*[1 << 30]C.uint32_t)(c.col)[: nnz+1 : nnz+1]
	va := (*[1 << 29]C.int64_t)(c.va)[: nnz+1 : nnz+1]
	k := 0
	for g, row := range m {
		rp[g] = C.uint32_t(k)
		for j, v := range row {
			if v != 0 {
				col[k] = C.uint32_t(j)
				va[k] = C.int64_t(v)
				k++
			}
		}
	}
	rp[len(m)] = C.uint32_t(k)
	c.csr.row_ptr = (*C.uint32_t)(c.rp)
	c.csr.col = (*C.uint32_t)(c.col)
	c.csr.val = (*C.int64_t)(c.va)
	return c
}

func (c *cCsr) free() {
	C.free(c.rp)
	C.free(c.col)
	C.free(c.va)
}

// HipQAP is the device-resident counterpart of ToQAP(circuit) (qap.go:35-65): the per-variable
// polynomials are never materialised, the quotient works from the sparse matrices on the domain {1..n}.
type HipQAP struct {
	h                     *C.ps_qap
	nbVars, nbIO, nbGates int
}

func NewHipQAP(circuit R1CS) *HipQAP {
	l, r, o := newCsr(circuit.left), newCsr(circuit.right), newCsr(circuit.out)
	defer l.free()
	defer r.free()
	defer o.free()
	q := &HipQAP{nbVars: len(circuit.vars), nbIO: circuit.nbIO(), nbGates: len(circuit.left)}
	check(C.ps_qap_create(hipCtx, C.size_t(q.nbGates), C.size_t(q.nbVars), C.size_t(q.nbIO), &l.csr, &r.csr, &o.csr, &q.h))
	return q
}

func (q *HipQAP) Free() { C.ps_qap_free(q.h) }

// QuotientHIP replaces `func (q QAP) Quotient(sol Vector) Poly` (qap.go:151-162): panics "apocalypse"
// when the witness does not satisfy the circuit.
func (q *HipQAP) QuotientHIP(sol Vector) Poly {
	dsol := uploadSolution(sol)
	defer C.ps_scalars_free(dsol)
	var h *C.ps_scalars
	check(C.ps_qap_quotient(hipCtx, q.h, dsol, nil, nil, nil, &h))
	defer C.ps_scalars_free(h)
	return downloadPoly(h)
}

// ---------------------------------------------------------------------------------------
// Groth16 (groth16.go)
// ---------------------------------------------------------------------------------------

// HipGroth16 holds what Groth16Prove / Groth16Verify need on the device: the CRS arrays of a
// Groth16Setup uploaded once, and the QAP.
type HipGroth16 struct {
	pk                        C.ps_groth16_pk
	vk                        C.ps_groth16_vk
	xi, xi2, nioLP, xiT, ioLP *C.ps_points
	lagrange                  []*C.ps_points // lxi, lxi2, lxi_t when the key came from NewHipGroth16FromToxicWaste
	qap                       *HipQAP
}

func NewHipGroth16(tr Groth16Setup, q *HipQAP) *HipGroth16 {
	hs := &HipGroth16{qap: q}
	hs.xi = uploadPoints(C.PS_G1, tr.Xi)
	hs.xi2 = uploadPoints(C.PS_G2, tr.Xi2) // declared []G1 at groth16.go:60, holds G2 points
	hs.nioLP = uploadPoints(C.PS_G1, tr.NioLP)
	hs.xiT = uploadPoints(C.PS_G1, tr.XiT)
	hs.ioLP = uploadPoints(C.PS_G1, tr.IoLP)
	copyTo(unsafe.Pointer(&hs.pk.alpha[0]), affineOf(C.PS_G1, tr.Alpha))
	copyTo(unsafe.Pointer(&hs.pk.beta[0]), affineOf(C.PS_G1, tr.Beta))
	copyTo(unsafe.Pointer(&hs.pk.delta[0]), affineOf(C.PS_G1, tr.Delta))
	copyTo(unsafe.Pointer(&hs.pk.beta2[0]), affineOf(C.PS_G2, tr.Beta2))
	copyTo(unsafe.Pointer(&hs.pk.delta2[0]), affineOf(C.PS_G2, tr.Delta2))
	hs.pk.xi, hs.pk.xi2, hs.pk.nio_lp, hs.pk.xi_t = hs.xi, hs.xi2, hs.nioLP, hs.xiT
	copyTo(unsafe.Pointer(&hs.vk.alpha[0]), affineOf(C.PS_G1, tr.Alpha))
	copyTo(unsafe.Pointer(&hs.vk.beta2[0]), affineOf(C.PS_G2, tr.Beta2))
	copyTo(unsafe.Pointer(&hs.vk.gamma[0]), affineOf(C.PS_G2, tr.Gamma))
	copyTo(unsafe.Pointer(&hs.vk.delta2[0]), affineOf(C.PS_G2, tr.Delta2))
	hs.vk.io_lp = hs.ioLP
	return hs
}

// NewHipGroth16FromToxicWaste builds the device-resident key from the five scalars the reference's setup keeps
// "for testing and learning purpose" (groth16.go:13-27, tr.tw): ps_groth16_setup recomputes every CRS array on the
// GPU -- the same points, byte for byte, as tr.Xi, tr.Xi2, tr.IoLP, tr.NioLP, tr.XiT -- and emits the SAME CRS in
// Lagrange form beside them (ps_groth16_pk.lxi / lxi2 / lxi_t), with which the prover needs no interpolation and no
// division (a 2^20-constraint proof in 21.5 ms instead of 32.5).  Nothing is uploaded through MarshalBinary.
func NewHipGroth16FromToxicWaste(tr Groth16Setup, q *HipQAP) *HipGroth16 {
	var tw C.ps_groth16_toxic
	put := func(dst *C.uint8_t, e Element) {
		b, err := e.MarshalBinary()
		if err != nil {
			panic(err)
		}
		copyTo(unsafe.Pointer(dst), b)
	}
	put(&tw.alpha[0], tr.tw.Alpha)
	put(&tw.beta[0], tr.tw.Beta)
	put(&tw.delta[0], tr.tw.Delta)
	put(&tw.x[0], tr.tw.X)
	put(&tw.gamma[0], tr.tw.Gamma)
	var crs C.ps_groth16_crs
	check(C.ps_groth16_setup(hipCtx, q.h, &tw, &crs))
	hs := &HipGroth16{qap: q, xi: crs.xi, xi2: crs.xi2, nioLP: crs.nio_lp, xiT: crs.xi_t, ioLP: crs.io_lp}
	hs.lagrange = []*C.ps_points{crs.lxi, crs.lxi2, crs.lxi_t}
	hs.pk.alpha, hs.pk.beta, hs.pk.delta, hs.pk.beta2, hs.pk.delta2 = crs.alpha, crs.beta, crs.delta, crs.beta2, crs.delta2
	hs.pk.xi, hs.pk.xi2, hs.pk.nio_lp, hs.pk.xi_t = crs.xi, crs.xi2, crs.nio_lp, crs.xi_t
	hs.pk.lxi, hs.pk.lxi2, hs.pk.lxi_t = crs.lxi, crs.lxi2, crs.lxi_t
	hs.vk.alpha, hs.vk.beta2, hs.vk.gamma, hs.vk.delta2 = crs.alpha, crs.beta2, crs.gamma, crs.delta2
	hs.vk.io_lp = crs.io_lp
	return hs
}

func (hs *HipGroth16) Free() {
	for _, p := range append([]*C.ps_points{hs.xi, hs.xi2, hs.nioLP, hs.xiT, hs.ioLP}, hs.lagrange...) {
		C.ps_points_free(p)
	}
}

// Groth16ProveHIP replaces `func Groth16Prove(tr Groth16Setup, q QAP, sol Vector) Groth16Proof`
// (groth16.go:122-211).  r and s are drawn exactly as the reference draws them (:148, :158) and
// handed to the library, which is deterministic.
func Groth16ProveHIP(hs *HipGroth16, sol Vector) Groth16Proof {
	r := NewElement().Pick(random.New())
	s := NewElement().Pick(random.New())
	rb, _ := r.MarshalBinary()
	sb, _ := s.MarshalBinary()
	dsol := uploadSolution(sol)
	defer C.ps_scalars_free(dsol)
	A := make([]byte, g1Wire)
	B := make([]byte, g2Wire)
	Cc := make([]byte, g1Wire)
	check(C.ps_groth16_prove(hipCtx, &hs.pk, hs.qap.h, dsol, u8(rb), u8(sb), u8(A), u8(B), u8(Cc)))
	return Groth16Proof{
		tp: groth16ToxicProof{R: r, S: s},
		A:  pointFrom(C.PS_G1, A, zeroG1),
		B:  pointFrom(C.PS_G2, B, zeroG2),
		C:  pointFrom(C.PS_G1, Cc, zeroG1),
	}
}

// Groth16VerifyHIP replaces `func Groth16Verify(tr Groth16Setup, q QAP, p Groth16Proof, io Vector) bool`
// (groth16.go:214-233): pairings on the host inside the library, the IO sum on the GPU.  A proof point
// that is not a canonical encoding of a subgroup element never gets here: UnmarshalBinary refused it.
func Groth16VerifyHIP(hs *HipGroth16, p Groth16Proof, io Vector) bool {
	dio := uploadSolution(io)
	defer C.ps_scalars_free(dio)
	var ok C.int
	check(C.ps_groth16_verify(hipCtx, &hs.vk, dio, u8(affineOf(C.PS_G1, p.A)), u8(affineOf(C.PS_G2, p.B)),
		u8(affineOf(C.PS_G1, p.C)), &ok))
	return ok != 0
}

// ---------------------------------------------------------------------------------------
// PHGR13 / Pinocchio (pinochio.go)
// ---------------------------------------------------------------------------------------

type HipPHGR13 struct {
	ek     C.ps_phgr13_ek
	vk     C.ps_phgr13_vk
	arrays []*C.ps_points
	qap    *HipQAP
}

func NewHipPHGR13(setup PHGR13Setup, q *HipQAP) *HipPHGR13 {
	hp := &HipPHGR13{qap: q}
	up := func(group C.int, pts []Commit) *C.ps_points {
		h := uploadPoints(group, pts)
		hp.arrays = append(hp.arrays, h)
		return h
	}
	ek := setup.EK
	hp.ek.vs, hp.ek.ws, hp.ek.ys = up(C.PS_G1, ek.vs), up(C.PS_G2, ek.ws), up(C.PS_G1, ek.ys)
	hp.ek.vas, hp.ek.was, hp.ek.yas = up(C.PS_G1, ek.vas), up(C.PS_G1, ek.was), up(C.PS_G1, ek.yas)
	hp.ek.gsi = up(C.PS_G1, ek.gsi)
	// wbs is declared []G2 (pinochio.go:60) but generated from g1w: G1 points (pinochio.go:136)
	hp.ek.vbs, hp.ek.wbs, hp.ek.ybs = up(C.PS_G1, ek.vbs), up(C.PS_G1, ek.wbs), up(C.PS_G1, ek.ybs)
	vk := setup.VK
	diff := q.nbVars - q.nbIO // the reference's split (pinochio.go:291)
	hp.vk.vs_io, hp.vk.ws_io, hp.vk.ys_io = up(C.PS_G1, vk.vs[:diff]), up(C.PS_G2, vk.ws[:diff]), up(C.PS_G1, vk.ys[:diff])
	copyTo(unsafe.Pointer(&hp.vk.av[0]), affineOf(C.PS_G2, vk.av))
	copyTo(unsafe.Pointer(&hp.vk.aw[0]), affineOf(C.PS_G1, vk.aw)) // declared G2, a G1 element (pinochio.go:146)
	copyTo(unsafe.Pointer(&hp.vk.ay[0]), affineOf(C.PS_G2, vk.ay))
	copyTo(unsafe.Pointer(&hp.vk.gamma[0]), affineOf(C.PS_G2, vk.gamma))
	copyTo(unsafe.Pointer(&hp.vk.bgamma[0]), affineOf(C.PS_G1, vk.bgamma))
	copyTo(unsafe.Pointer(&hp.vk.bgamma2[0]), affineOf(C.PS_G2, vk.bgamma2))
	copyTo(unsafe.Pointer(&hp.vk.yts[0]), affineOf(C.PS_G2, vk.yts))
	return hp
}

func (hp *HipPHGR13) Free() {
	for _, p := range hp.arrays {
		C.ps_points_free(p)
	}
}

func bytesOf(p unsafe.Pointer, n int) []byte { return C.GoBytes(p, C.int(n)) }

// PHGR13ProveHIP replaces `func PHGR13Prove(ek PHGR13EvalKey, qap QAP, solution Vector) PHGR13Proof`
// (pinochio.go:207-254).  Deterministic.
func PHGR13ProveHIP(hp *HipPHGR13, solution Vector) PHGR13Proof {
	dsol := uploadSolution(solution)
	defer C.ps_scalars_free(dsol)
	var out C.ps_phgr13_proof
	check(C.ps_phgr13_prove(hipCtx, &hp.ek, hp.qap.h, dsol, &out))
	g1 := func(p *C.uint8_t) Commit { return pointFrom(C.PS_G1, bytesOf(unsafe.Pointer(p), g1Wire), zeroG1) }
	return PHGR13Proof{
		vss:  g1(&out.vss[0]),
		vass: g1(&out.vass[0]),
		wss:  pointFrom(C.PS_G2, bytesOf(unsafe.Pointer(&out.wss[0]), g2Wire), zeroG2),
		wass: g1(&out.wass[0]),
		yss:  g1(&out.yss[0]),
		yass: g1(&out.yass[0]),
		hs:   g1(&out.hs[0]),
		gz:   g1(&out.gz[0]),
	}
}

// PHGR13VerifyHIP replaces `func PHGR13Verify(vk PHGR13VerifKey, qap QAP, p PHGR13Proof, io Vector) bool`
// (pinochio.go:281-378).
func PHGR13VerifyHIP(hp *HipPHGR13, p PHGR13Proof, io Vector) bool {
	var in C.ps_phgr13_proof
	copyTo(unsafe.Pointer(&in.vss[0]), affineOf(C.PS_G1, p.vss))
	copyTo(unsafe.Pointer(&in.vass[0]), affineOf(C.PS_G1, p.vass))
	copyTo(unsafe.Pointer(&in.wss[0]), affineOf(C.PS_G2, p.wss))
	copyTo(unsafe.Pointer(&in.wass[0]), affineOf(C.PS_G1, p.wass))
	copyTo(unsafe.Pointer(&in.yss[0]), affineOf(C.PS_G1, p.yss))
	copyTo(unsafe.Pointer(&in.yass[0]), affineOf(C.PS_G1, p.yass))
	copyTo(unsafe.Pointer(&in.hs[0]), affineOf(C.PS_G1, p.hs))
	copyTo(unsafe.Pointer(&in.gz[0]), affineOf(C.PS_G1, p.gz))
	dio := uploadSolution(io)
	defer C.ps_scalars_free(dio)
	var ok C.int
	check(C.ps_phgr13_verify(hipCtx, &hp.vk, dio, &in, &ok))
	return ok != 0
}

// ---------------------------------------------------------------------------------------
// the reference's own tests, pointed at the GPU backend (groth16_test.go:22-30, pinocchio_test.go:23-29):
//
//	r1cs := createR1CS(); s := createWitness(r1cs); qap := ToQAP(r1cs)
//	hq := NewHipQAP(r1cs)
//	tr := NewGroth16TrustedSetup(qap); hs := NewHipGroth16(tr, hq)
//	proof := Groth16ProveHIP(hs, s)
//	require.True(t, Groth16Verify(tr, qap, proof, s[:qap.nbVars-qap.nbIO]))     // the reference's CPU verifier
//	require.True(t, Groth16VerifyHIP(hs, proof, s[:qap.nbVars-qap.nbIO]))      // or the library's
//	setup := NewPHGR13TrustedSetup(qap); hp := NewHipPHGR13(setup, hq)
//	require.True(t, PHGR13Verify(setup.VK, qap, PHGR13ProveHIP(hp, s), s[:qap.nbVars-qap.nbIO]))
// ---------------------------------------------------------------------------------------
