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
// 8 row f4) and has never been compiled or vetted: do that first (`go vet -tags hip`) wherever Go is available.
// Every C entry point it calls -- the multi-device ones included -- is exercised by tests/abi_smoke.c (plain C) and by
// the ctypes binding the test-suite runs on.
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
	"runtime"
	"sync"
	"unsafe"

	"github.com/drand/kyber/util/random"
)

// ---------------------------------------------------------------------------------------
// context and error mapping
// ---------------------------------------------------------------------------------------

// The reference's Groth16Prove / PHGR13Prove / BlindEval are pure functions that any number of goroutines may call.
// A ps_ctx is NOT thread-safe (one stream, one workspace, one queue of pending sums), so every use of the package
// context is serialised by hipMu.  (A pool of contexts -- one per caller, all over the same uploaded key arrays, which
// the library allows: include/playsnark_hip.h "Thread safety" -- is the way to prove from several goroutines at once.)
var (
	hipCtx *C.ps_ctx
	hipMu  sync.Mutex
)

func init() {
	if v := int(C.ps_abi_version()); v != C.PS_ABI_VERSION {
		panic(fmt.Sprintf("playsnark_hip: library ABI %d, shim written for %d", v, C.PS_ABI_VERSION))
	}
	runtime.LockOSThread() // ps_last_error is thread-local: read it on the thread that made the failing call
	defer runtime.UnlockOSThread()
	if rc := C.ps_ctx_create(0, &hipCtx); rc != C.PS_OK {
		// no CPU fallback behind this build tag: fail as loudly as the reference's own panics
		panic("playsnark_hip: " + C.GoString(C.ps_last_error()))
	}
}

// call runs one library call that uses hipCtx: under the context's lock, and pinned to its OS thread so that
// ps_last_error() (thread-local in the library) is read on the thread the failing call ran on -- a goroutine may
// otherwise be rescheduled onto another thread between two cgo calls.  Error codes map onto the reference's panics.
func call(f func() C.int) {
	hipMu.Lock()
	runtime.LockOSThread()
	rc := f()
	msg := ""
	if rc != C.PS_OK {
		msg = C.GoString(C.ps_last_error())
	}
	runtime.UnlockOSThread()
	hipMu.Unlock()
	switch rc {
	case C.PS_OK:
	case C.PS_ERR_NOT_DIVISIBLE:
		panic("apocalypse") // qap.go:159, pinochio.go:215
	default:
		// PS_ERR_LENGTH carries the message of algebra.go:351, PS_ERR_ARG that of sanityCheck (qap.go:177-189)
		panic(msg)
	}
}

// check is `call` for the entry points that touch no context (ps_point_convert, ps_points_sum, ..).
func check(f func() C.int) {
	runtime.LockOSThread()
	rc := f()
	msg := ""
	if rc != C.PS_OK {
		msg = C.GoString(C.ps_last_error())
	}
	runtime.UnlockOSThread()
	if rc != C.PS_OK {
		panic(msg)
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
	check(func() C.int { return C.ps_point_convert(group, C.PS_FMT_COMPRESSED, C.PS_FMT_AFFINE, u8(in), u8(out)) })
	return out
}

// pointFrom turns an affine result of the library back into a kyber point of the group of `like`.
func pointFrom(group C.int, affine []byte, like Commit) Commit {
	comp := make([]byte, wireLen(group)/2)
	check(func() C.int { return C.ps_point_convert(group, C.PS_FMT_AFFINE, C.PS_FMT_COMPRESSED, u8(affine), u8(comp)) })
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
	call(func() C.int { return C.ps_points_upload(hipCtx, group, u8(raw), C.size_t(len(pts)), C.PS_FMT_COMPRESSED, &h) })
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
	call(func() C.int { return C.ps_scalars_upload_i64(hipCtx, p, C.size_t(len(vals)), &h) })
	return h
}

func downloadPoly(h *C.ps_scalars) Poly {
	n := int(C.ps_scalars_len(h))
	raw := make([]byte, 32*n)
	call(func() C.int { return C.ps_scalars_download(hipCtx, h, 0, C.size_t(n), u8(raw)) })
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
	call(func() C.int { return C.ps_msm_be32(hipCtx, crs, u8(sc), C.size_t(len(p)), u8(out)) })
	return pointFrom(group, out, zero)
}

// MulHIP replaces `func (p Poly) Mul(p2 Poly) Poly`.
func (p Poly) MulHIP(p2 Poly) Poly {
	var a, b, prod *C.ps_scalars
	ra, rb := marshalScalars(p), marshalScalars(p2)
	call(func() C.int { return C.ps_scalars_upload(hipCtx, u8(ra), C.size_t(len(p)), &a) })
	defer C.ps_scalars_free(a)
	call(func() C.int { return C.ps_scalars_upload(hipCtx, u8(rb), C.size_t(len(p2)), &b) })
	defer C.ps_scalars_free(b)
	call(func() C.int { return C.ps_poly_mul(hipCtx, a, b, &prod) })
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
	call(func() C.int { return C.ps_qap_create(hipCtx, C.size_t(q.nbGates), C.size_t(q.nbVars), C.size_t(q.nbIO), &l.csr, &r.csr, &o.csr, &q.h) })
	return q
}

func (q *HipQAP) Free() { C.ps_qap_free(q.h) }

// QuotientHIP replaces `func (q QAP) Quotient(sol Vector) Poly` (qap.go:151-162): panics "apocalypse"
// when the witness does not satisfy the circuit.
func (q *HipQAP) QuotientHIP(sol Vector) Poly {
	dsol := uploadSolution(sol)
	defer C.ps_scalars_free(dsol)
	var h *C.ps_scalars
	call(func() C.int { return C.ps_qap_quotient(hipCtx, q.h, dsol, nil, nil, nil, &h) })
	defer C.ps_scalars_free(h)
	return downloadPoly(h)
}

// IsValidHIP replaces `func (q *QAP) IsValid(sol Vector) bool` (qap.go:107-148): does z(x) divide
// left(x) right(x) - out(x)?  On the device: the three SpMVs and the gate check of the quotient.
func (q *HipQAP) IsValidHIP(sol Vector) bool {
	dsol := uploadSolution(sol)
	defer C.ps_scalars_free(dsol)
	var ok C.int
	call(func() C.int { return C.ps_qap_is_valid(hipCtx, q.h, dsol, &ok) })
	return ok != 0
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

// ToLagrange puts a key made by the reference's NewGroth16TrustedSetup (monomial arrays only) onto the prover's fast route
// WITHOUT the toxic waste, which "must be delete[d] after a trusted setup" (groth16.go:13-14): Xi, Xi2 and XiT are converted
// on the GPU into l_j(x) G1, l_j(x) G2 and lambda_k(x) t(x)/delta G1 over the group elements alone
// (ps_points_monomial_to_lagrange: a transposed interpolation, seconds at 2^16 gates, under a minute at 2^20 -- once per key).
// Afterwards Groth16ProveHIP needs no interpolation and no division; the proof bytes do not change.
func (hs *HipGroth16) ToLagrange() {
	if hs.pk.lxi != nil {
		return
	}
	var lxi, lxi2, lxit *C.ps_points
	call(func() C.int { return C.ps_points_monomial_to_lagrange(hipCtx, hs.qap.h, hs.xi, 0, &lxi) })
	call(func() C.int { return C.ps_points_monomial_to_lagrange(hipCtx, hs.qap.h, hs.xi2, 0, &lxi2) })
	call(func() C.int { return C.ps_points_monomial_to_lagrange(hipCtx, hs.qap.h, hs.xiT, 1, &lxit) })
	hs.lagrange = []*C.ps_points{lxi, lxi2, lxit}
	hs.pk.lxi, hs.pk.lxi2, hs.pk.lxi_t = lxi, lxi2, lxit
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
	call(func() C.int { return C.ps_groth16_setup(hipCtx, q.h, &tw, &crs) })
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
	call(func() C.int { return C.ps_groth16_prove(hipCtx, &hs.pk, hs.qap.h, dsol, u8(rb), u8(sb), u8(A), u8(B), u8(Cc)) })
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
	call(func() C.int { return C.ps_groth16_verify(hipCtx, &hs.vk, dio, u8(affineOf(C.PS_G1, p.A)), u8(affineOf(C.PS_G2, p.B)),
		u8(affineOf(C.PS_G1, p.C)), &ok) })
	return ok != 0
}

// ---------------------------------------------------------------------------------------
// Several GPUs from ONE process (a cgo caller cannot wrap a function call in one process per GPU, which is how
// bench.py and playsnark_amd/dist.py scale): ps_msm_multi_device and ps_groth16_prove_multi run one context per
// device side by side and fold the per-device partial sums on the host (SURVEY.md 8e, DESIGN.md section 7).
// ---------------------------------------------------------------------------------------

// shardRange is the index range [first, first+count) of device d of ndev: contiguous, sizes differ by at most one
// (the split ps_groth16_prove_multi checks with PS_ERR_LENGTH).
func shardRange(n, d, ndev int) (int, int) {
	base, extra := n/ndev, n%ndev
	first := d*base + minInt(d, extra)
	if d < extra {
		return first, base + 1
	}
	return first, base
}

func minInt(a, b int) int {
	if a < b {
		return a
	}
	return b
}

// HipDevices is one context per GPU of this process.
type HipDevices struct {
	mu   sync.Mutex
	ctxs []*C.ps_ctx
}

func NewHipDevices(ndev int) *HipDevices {
	hd := &HipDevices{}
	for d := 0; d < ndev; d++ {
		var c *C.ps_ctx
		check(func() C.int { return C.ps_ctx_create(C.int(d), &c) })
		hd.ctxs = append(hd.ctxs, c)
	}
	return hd
}

func (hd *HipDevices) Free() {
	for _, c := range hd.ctxs {
		C.ps_ctx_destroy(c)
	}
}

// HipShardedPoints is a CRS slice cut by index range over the devices (device d holds shardRange(len, d, ndev)).
type HipShardedPoints struct {
	parts []*C.ps_points
	n     int
}

func (hd *HipDevices) UploadPoints(group C.int, pts []Commit) *HipShardedPoints {
	sp := &HipShardedPoints{n: len(pts)}
	for d, c := range hd.ctxs {
		first, cnt := shardRange(len(pts), d, len(hd.ctxs))
		raw := marshalPoints(pts[first : first+cnt])
		var h *C.ps_points
		cc := c
		check(func() C.int { return C.ps_points_upload(cc, group, u8(raw), C.size_t(cnt), C.PS_FMT_COMPRESSED, &h) })
		sp.parts = append(sp.parts, h)
	}
	return sp
}

func (sp *HipShardedPoints) Free() {
	for _, p := range sp.parts {
		C.ps_points_free(p)
	}
}

// BlindEvalHIPMulti is Poly.BlindEval (algebra.go:348-359) with the sum sharded over the devices.
func (p Poly) BlindEvalHIPMulti(zero Commit, hd *HipDevices, crs *HipShardedPoints) Commit {
	if crs.n != len(p) { // algebra.go:350-352
		panic(fmt.Sprintf("mismatch of length between poly %d and blinded eval points %d", len(p), crs.n))
	}
	hd.mu.Lock()
	defer hd.mu.Unlock()
	ndev := len(hd.ctxs)
	scal := make([]*C.ps_scalars, ndev)
	for d, c := range hd.ctxs {
		first, cnt := shardRange(len(p), d, ndev)
		raw := marshalScalars(p[first : first+cnt])
		cc, dd := c, d
		check(func() C.int { return C.ps_scalars_upload(cc, u8(raw), C.size_t(cnt), &scal[dd]) })
		defer C.ps_scalars_free(scal[d])
	}
	group := C.ps_points_group(crs.parts[0])
	out := make([]byte, wireLen(group))
	// the three pointer arrays live in C memory (cgo: no Go pointers to Go pointers across the boundary)
	arr := func(n int) unsafe.Pointer { return C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))) }
	cctx, cpts, csc := arr(ndev), arr(ndev), arr(ndev)
	defer C.free(cctx)
	defer C.free(cpts)
	defer C.free(csc)
	for d := 0; d < ndev; d++ {
		(*[1 << 20]*C.ps_ctx)(cctx)[d] = hd.ctxs[d]
		(*[1 << 20]*C.ps_points)(cpts)[d] = crs.parts[d]
		(*[1 << 20]*C.ps_scalars)(csc)[d] = scal[d]
	}
	check(func() C.int {
		return C.ps_msm_multi_device((**C.ps_ctx)(cctx), (**C.ps_points)(cpts), (**C.ps_scalars)(csc), C.size_t(ndev), u8(out))
	})
	return pointFrom(group, out, zero)
}

// HipGroth16Multi: every device holds only ITS index range of Xi, Xi2, NioLP, XiT (an eighth of the key on eight GPUs),
// its own QAP of the circuit and its own copy of the solution.
type HipGroth16Multi struct {
	hd     *HipDevices
	dev    []C.ps_groth16_device
	qaps   []*C.ps_qap
	arrays []*HipShardedPoints
}

func NewHipGroth16Multi(hd *HipDevices, tr Groth16Setup, circuit R1CS) *HipGroth16Multi {
	hm := &HipGroth16Multi{hd: hd, dev: make([]C.ps_groth16_device, len(hd.ctxs))} // zero-initialised, as the header requires
	xi, xi2 := hd.UploadPoints(C.PS_G1, tr.Xi), hd.UploadPoints(C.PS_G2, tr.Xi2)
	nio, xit := hd.UploadPoints(C.PS_G1, tr.NioLP), hd.UploadPoints(C.PS_G1, tr.XiT)
	hm.arrays = []*HipShardedPoints{xi, xi2, nio, xit}
	l, r, o := newCsr(circuit.left), newCsr(circuit.right), newCsr(circuit.out)
	defer l.free()
	defer r.free()
	defer o.free()
	// check() panics on a library error: a device that fails after others succeeded must not leak their QAPs and shards
	done := false
	defer func() {
		if !done {
			hm.Free()
		}
	}()
	for d, c := range hd.ctxs {
		var q *C.ps_qap
		cc := c
		check(func() C.int {
			return C.ps_qap_create(cc, C.size_t(len(circuit.left)), C.size_t(len(circuit.vars)), C.size_t(circuit.nbIO()), &l.csr, &r.csr, &o.csr, &q)
		})
		hm.qaps = append(hm.qaps, q)
		hm.dev[d].ctx, hm.dev[d].qap = c, q
		pk := &hm.dev[d].pk
		pk.xi, pk.xi2, pk.nio_lp, pk.xi_t = xi.parts[d], xi2.parts[d], nio.parts[d], xit.parts[d]
		if d == 0 { // the fixed points are read from dev[0].pk
			copyTo(unsafe.Pointer(&pk.alpha[0]), affineOf(C.PS_G1, tr.Alpha))
			copyTo(unsafe.Pointer(&pk.beta[0]), affineOf(C.PS_G1, tr.Beta))
			copyTo(unsafe.Pointer(&pk.delta[0]), affineOf(C.PS_G1, tr.Delta))
			copyTo(unsafe.Pointer(&pk.beta2[0]), affineOf(C.PS_G2, tr.Beta2))
			copyTo(unsafe.Pointer(&pk.delta2[0]), affineOf(C.PS_G2, tr.Delta2))
		}
	}
	done = true
	return hm
}

func (hm *HipGroth16Multi) Free() {
	for _, q := range hm.qaps {
		C.ps_qap_free(q)
	}
	for _, a := range hm.arrays {
		a.Free()
	}
}

// Groth16ProveHIPMulti is Groth16Prove (groth16.go:122-211) over the devices of this process: same proof bytes as
// Groth16ProveHIP for the same (r, s).
func Groth16ProveHIPMulti(hm *HipGroth16Multi, sol Vector) Groth16Proof {
	r := NewElement().Pick(random.New())
	s := NewElement().Pick(random.New())
	rb, _ := r.MarshalBinary()
	sb, _ := s.MarshalBinary()
	hm.hd.mu.Lock()
	defer hm.hd.mu.Unlock()
	vals := make([]C.int64_t, len(sol))
	for i, v := range sol {
		vals[i] = C.int64_t(v)
	}
	// the device structs go to C memory: they hold C pointers only, but live in a Go slice
	cdev := (*C.ps_groth16_device)(C.malloc(C.size_t(len(hm.dev)) * C.size_t(unsafe.Sizeof(hm.dev[0]))))
	defer C.free(unsafe.Pointer(cdev))
	devs := (*[1 << 16]C.ps_groth16_device)(unsafe.Pointer(cdev))[:len(hm.dev):len(hm.dev)]
	var vp *C.int64_t // an empty solution has no first element to point at (the library refuses it by length, not by a Go panic here)
	if len(vals) > 0 {
		vp = &vals[0]
	}
	for d := range hm.dev {
		var h *C.ps_scalars
		cc := hm.dev[d].ctx
		check(func() C.int { return C.ps_scalars_upload_i64(cc, vp, C.size_t(len(vals)), &h) })
		defer C.ps_scalars_free(h)
		devs[d] = hm.dev[d]
		devs[d].sol = h
	}
	A, B, Cc := make([]byte, g1Wire), make([]byte, g2Wire), make([]byte, g1Wire)
	check(func() C.int {
		return C.ps_groth16_prove_multi(cdev, C.size_t(len(hm.dev)), u8(rb), u8(sb), u8(A), u8(B), u8(Cc))
	})
	return Groth16Proof{
		tp: groth16ToxicProof{R: r, S: s},
		A:  pointFrom(C.PS_G1, A, zeroG1),
		B:  pointFrom(C.PS_G2, B, zeroG2),
		C:  pointFrom(C.PS_G1, Cc, zeroG1),
	}
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
	call(func() C.int { return C.ps_phgr13_prove(hipCtx, &hp.ek, hp.qap.h, dsol, &out) })
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
	call(func() C.int { return C.ps_phgr13_verify(hipCtx, &hp.vk, dio, &in, &ok) })
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
//	require.True(t, hq.IsValidHIP(s))                                            // TestQAPValidity, qap_test.go
//	hd := NewHipDevices(8); hm := NewHipGroth16Multi(hd, tr, r1cs)              // one process, eight GPUs
//	require.True(t, Groth16Verify(tr, qap, Groth16ProveHIPMulti(hm, s), s[:qap.nbVars-qap.nbIO]))
// ---------------------------------------------------------------------------------------
