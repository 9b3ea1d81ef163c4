//go:build hip

// Package hip is the cgo binding of libsfgwas_hip.so (include/sfgwas_hip.h) for hhcho/sfgwas.
//
// Where it goes: copy this directory to github.com/hhcho/sfgwas/hip and the sibling files to gwas/, crypto/ and mpc/
// (each carries the build tag `hip`; the files they replace get `//go:build !hip`, see INTEGRATION.md §0).
//
// NOT COMPILED IN THE sfgwas-hip REPOSITORY: its image has no Go toolchain and the lattigo fork / mpc-core modules are not
// vendored in the reference tree.  The lattigo API used here is the one the reference itself uses (ckks.Ciphertext.Value()[k].Coeffs[l],
// ckks.NewCiphertext, ckks.RotationKeySet.Keys, ckks.SwitchingKey.Value, ring.NewRing / PsiMont / MredParams / InvMForm); a maintainer
// should expect to fix small spelling differences against the fork on the first build.
//
// This package imports lattigo only (never sfgwas/crypto, gwas or mpc), so all three can import it without a cycle.
package hip

/*
#cgo CFLAGS: -I${SRCDIR}/../third_party/sfgwas-hip/include
#cgo LDFLAGS: -L${SRCDIR}/../third_party/sfgwas-hip/sfgwas_amd/lib -lsfgwas_hip -Wl,-rpath,${SRCDIR}/../third_party/sfgwas-hip/sfgwas_amd/lib
#include <stdlib.h>
#include "sfgwas_hip.h"
*/
import "C"

import (
	"fmt"
	"sync"
	"unsafe"

	"github.com/ldsec/lattigo/v2/ckks"
	"github.com/ldsec/lattigo/v2/ring"
)

// Ctx is one caller's handle on the library: tables and keys are shared between forks, queues and scratch are per handle.
// The reference runs MatMult4Stream from several goroutines at once (gwas/assoc.go:360-408): each takes its own Fork().
type Ctx struct {
	p      *C.sfg_ctx
	mg     *C.sfg_mgpu // non-nil when Init was given several devices: the products and the association scan run on all of them (include/sfgwas_hip.h, "SURVEY 8e")
	Params *ckks.Parameters
	N      int // ring degree
	NQ, NP int // moduli of Q and of P
}

// Default is the context the drop-in function bodies use; Init sets it (gwas/gwas.go:212, after CollectiveInit).
var Default *Ctx

// check converts a non-zero return code into a panic: the reference panics / log.Fatals on this path
// (gwas/matmult.go:361, gwas/filestream.go:60,334, crypto/basics.go:817), it never returns errors.
func (h *Ctx) check(rc C.int, what string) {
	if rc != 0 {
		panic(fmt.Sprintf("sfgwas-hip: %s: %s", what, C.GoString(C.sfg_last_error(h.p))))
	}
}

// Init creates the device context(s) from the CKKS parameters and uploads every switching key the party holds.
//   rotKs: cryptoParams.RotKs (crypto/crypto.go:50, filled at :208), rlk: cryptoParams.Rlk (:49).
//   devices: the HIP devices of this party's node.  One device: a plain context.  Several: the multi-GPU engine (sfg_mgpu_create: one context per device,
//   RCCL inside the library) - the genotype matrix registered by MatMult4StreamPreprocess is sharded by SNP block over them, MatMult4StreamCompute and the
//   association scan run on all of them, and everything else (evaluator ops, per-call MatMult4Stream forks) runs on devices[0]'s context.
func Init(params *ckks.Parameters, rotKs *ckks.RotationKeySet, rlk *ckks.RelinearizationKey, devices []int) *Ctx {
	if len(devices) == 0 {
		devices = []int{0}
	}
	qi, pi := params.Qi(), params.Pi()
	moduli := append(append([]uint64{}, qi...), pi...)
	ringQP, err := ring.NewRing(params.N(), moduli)
	if err != nil {
		panic(err)
	}
	// lattigo's own primitive 2N-th roots, out of Montgomery form: the NTT output order then is lattigo's, whichever root its
	// parameter generation picked (include/sfgwas_hip.h: sfg_ctx_create, `psi`)
	psi := make([]uint64, len(moduli))
	for i := range moduli {
		psi[i] = ring.InvMForm(ringQP.PsiMont[i], moduli[i], ringQP.MredParams[i])
	}
	h := &Ctx{Params: params, N: params.N(), NQ: len(qi), NP: len(pi)}
	if len(devices) == 1 {
		var c *C.sfg_ctx
		if C.sfg_ctx_create(&c, C.int(devices[0]), C.int(params.LogN()), C.int(len(qi)), C.int(len(pi)),
			(*C.uint64_t)(unsafe.Pointer(&moduli[0])), (*C.uint64_t)(unsafe.Pointer(&psi[0])), C.double(params.Scale())) != 0 {
			panic("sfgwas-hip: sfg_ctx_create: " + C.GoString(C.sfg_last_error(nil)))
		}
		h.p = c
	} else {
		devs := make([]C.int, len(devices))
		for i, d := range devices {
			devs[i] = C.int(d)
		}
		var m *C.sfg_mgpu
		if C.sfg_mgpu_create(&m, &devs[0], C.int(len(devs)), C.int(params.LogN()), C.int(len(qi)), C.int(len(pi)),
			(*C.uint64_t)(unsafe.Pointer(&moduli[0])), (*C.uint64_t)(unsafe.Pointer(&psi[0])), C.double(params.Scale())) != 0 {
			panic("sfgwas-hip: sfg_mgpu_create: " + C.GoString(C.sfg_mgpu_last_error(nil)))
		}
		h.mg = m
		h.p = C.sfg_mgpu_ctx(m, 0)
	}
	if rotKs != nil {
		for galEl, swk := range rotKs.Keys {
			flat := FlattenSwitchingKey(swk, h.NQ+h.NP, h.N)
			if h.mg != nil {
				h.mcheck(C.sfg_mgpu_load_rotkey(h.mg, C.uint64_t(galEl), (*C.uint64_t)(unsafe.Pointer(&flat[0])), 1), "mgpu_load_rotkey")
			} else {
				h.check(C.sfg_ctx_load_rotkey(h.p, C.uint64_t(galEl), (*C.uint64_t)(unsafe.Pointer(&flat[0])), 1), "load_rotkey")
			}
		}
	}
	if rlk != nil && len(rlk.Keys) > 0 {
		flat := FlattenSwitchingKey(rlk.Keys[0], h.NQ+h.NP, h.N)
		if h.mg != nil {
			h.mcheck(C.sfg_mgpu_load_relinkey(h.mg, (*C.uint64_t)(unsafe.Pointer(&flat[0])), 1), "mgpu_load_relinkey")
		} else {
			h.check(C.sfg_ctx_load_relinkey(h.p, (*C.uint64_t)(unsafe.Pointer(&flat[0])), 1), "load_relinkey")
		}
	}
	Default = h
	return h
}

// mcheck is check for the multi-GPU engine's calls (the failing rank is named in the message).
func (h *Ctx) mcheck(rc C.int, what string) {
	if rc != 0 {
		panic(fmt.Sprintf("sfgwas-hip: %s: %s", what, C.GoString(C.sfg_mgpu_last_error(h.mg))))
	}
}

// Fork returns a handle for another goroutine: same keys and tables, own queues (sfg_ctx_fork).
func (h *Ctx) Fork() *Ctx {
	var c *C.sfg_ctx
	h.check(C.sfg_ctx_fork(h.p, &c), "fork")
	f := *h
	f.p = c
	f.mg = nil // a fork is one caller on devices[0]; the multi-GPU calls belong to the root handle
	return &f
}

// Close destroys the handle (a fork: its queues and scratch; the root: everything, on every device).
func (h *Ctx) Close() {
	if h.mg != nil {
		C.sfg_mgpu_destroy(h.mg) // owns the per-device contexts, h.p among them
		h.mg, h.p = nil, nil
		return
	}
	C.sfg_ctx_destroy(h.p)
	h.p = nil
}

// Raw exposes the C handle to the sibling files of this binding.
func (h *Ctx) Raw() unsafe.Pointer { return unsafe.Pointer(h.p) }

// ----------------------------------------------------------------------------------------------------------------
// Flat layouts.  cgo may read Go memory only for the duration of a call and only if it holds no Go pointers, so the
// [][]uint64 of lattigo polynomials are copied into one []uint64 per call (SURVEY.md §8b "Ownership").

// FlattenSwitchingKey: swk.Value[i][k].Coeffs[m][:] -> [beta][2][nq+np][N] (lattigo keeps these rows in NTT + Montgomery form).
func FlattenSwitchingKey(swk *ckks.SwitchingKey, nmod, n int) []uint64 {
	beta := len(swk.Value)
	flat := make([]uint64, beta*2*nmod*n)
	for i := 0; i < beta; i++ {
		for k := 0; k < 2; k++ {
			for m := 0; m < nmod; m++ {
				copy(flat[((i*2+k)*nmod+m)*n:], swk.Value[i][k].Coeffs[m][:n])
			}
		}
	}
	return flat
}

// CtWords is the number of uint64 words of a degree-1 ciphertext at `level`.
func (h *Ctx) CtWords(level int) int { return 2 * (level + 1) * h.N }

// FlattenCt writes ct.Value()[k].Coeffs[l] for l <= level into dst ([2][level+1][N]); the ciphertext must be at a level >= `level`
// (rows above `level` are dropped: lattigo's DropLevel keeps the first rows).
func (h *Ctx) FlattenCt(ct *ckks.Ciphertext, level int, dst []uint64) {
	nl := level + 1
	for k := 0; k < 2; k++ {
		for l := 0; l < nl; l++ {
			copy(dst[(k*nl+l)*h.N:], ct.Value()[k].Coeffs[l][:h.N])
		}
	}
}

// FlattenVec: a CipherVector ([]*ckks.Ciphertext) at one level, with length-1 broadcasting to n entries.
func (h *Ctx) FlattenVec(v []*ckks.Ciphertext, n, level int) []uint64 {
	w := h.CtWords(level)
	flat := make([]uint64, n*w)
	for i := 0; i < n; i++ {
		src := v[0]
		if len(v) > 1 {
			src = v[i]
		}
		h.FlattenCt(src, level, flat[i*w:(i+1)*w])
	}
	return flat
}

// FlattenCipherMatrix: A[i][b] -> [s][nbr][2][level+1][N].
func (h *Ctx) FlattenCipherMatrix(A [][]*ckks.Ciphertext, level int) []uint64 {
	s, nbr, w := len(A), len(A[0]), h.CtWords(level)
	flat := make([]uint64, s*nbr*w)
	for i := 0; i < s; i++ {
		for b := 0; b < nbr; b++ {
			h.FlattenCt(A[i][b], level, flat[(i*nbr+b)*w:(i*nbr+b+1)*w])
		}
	}
	return flat
}

// CtFromFlat builds a fresh ckks.Ciphertext (the callers mutate results in place, e.g. eval.Sub(out, .., out) at gwas/matmult.go:56,
// so results must be new objects) from [2][level+1][N] words.
func (h *Ctx) CtFromFlat(src []uint64, level int, scale float64) *ckks.Ciphertext {
	ct := ckks.NewCiphertext(h.Params, 1, level, scale)
	nl := level + 1
	for k := 0; k < 2; k++ {
		for l := 0; l < nl; l++ {
			copy(ct.Value()[k].Coeffs[l], src[(k*nl+l)*h.N:(k*nl+l+1)*h.N])
		}
	}
	return ct
}

// VecFromFlat: n ciphertexts from consecutive flat blocks.
func (h *Ctx) VecFromFlat(src []uint64, n, level int, scale float64) []*ckks.Ciphertext {
	w := h.CtWords(level)
	out := make([]*ckks.Ciphertext, n)
	for i := range out {
		out[i] = h.CtFromFlat(src[i*w:(i+1)*w], level, scale)
	}
	return out
}

// AddFlatInto: out[i][j] += ciphertext(flat[i][j]) with lattigo's own Add.  MatMult4Stream / ...Compute start from crypto.CZeroMat - a FRESH
// ENCRYPTION of zero (gwas/matmult.go:1174,1225,1443; crypto/basics.go:367-384) - and add the deterministic sum onto it; the shim
// keeps that, so outputs carry the same fresh randomness the reference's do.
func (h *Ctx) AddFlatInto(eval ckks.Evaluator, out [][]*ckks.Ciphertext, flat []uint64, level int, scale float64) {
	w := h.CtWords(level)
	for i := range out {
		for j := range out[i] {
			k := i*len(out[i]) + j
			ct := h.CtFromFlat(flat[k*w:(k+1)*w], level, scale)
			out[i][j].SetScale(scale) // CZeroMat encrypts at the default scale; the sum carries A.scale * Params.Scale (matmult.go:1045)
			if out[i][j].Level() > level {
				eval.DropLevel(out[i][j], out[i][j].Level()-level)
			}
			eval.Add(out[i][j], ct, out[i][j])
		}
	}
}

// ----------------------------------------------------------------------------------------------------------------
// Device buffers (uint64 words unless stated).

type DevBuf struct {
	h     *Ctx
	p     unsafe.Pointer
	Bytes int
}

func (h *Ctx) Alloc(bytes int) *DevBuf {
	var p unsafe.Pointer
	h.check(C.sfg_malloc(h.p, &p, C.size_t(bytes)), "malloc")
	return &DevBuf{h, p, bytes}
}
func (b *DevBuf) Free()               { b.h.check(C.sfg_free(b.h.p, b.p), "free"); b.p = nil }
func (b *DevBuf) Ptr() unsafe.Pointer { return b.p }
func (b *DevBuf) U64() *C.uint64_t    { return (*C.uint64_t)(b.p) }

// Upload copies host words to a new device buffer.
func (h *Ctx) Upload(words []uint64) *DevBuf {
	b := h.Alloc(8 * len(words))
	h.check(C.sfg_memcpy_h2d(h.p, b.p, unsafe.Pointer(&words[0]), C.size_t(8*len(words))), "h2d")
	return b
}

// Download copies a device buffer back (synchronises the handle's queue; fails while an unprovable encoder rounding is outstanding).
func (b *DevBuf) Download() []uint64 {
	out := make([]uint64, b.Bytes/8)
	b.h.check(C.sfg_memcpy_d2h(b.h.p, unsafe.Pointer(&out[0]), b.p, C.size_t(b.Bytes)), "d2h")
	return out
}

// ----------------------------------------------------------------------------------------------------------------
// Resident genotype matrices, keyed by the reference's cache-file prefix (gwas/pca.go:112-113 passes one prefix for X and one
// for X^T: ONE resident int8 copy serves both, the second prefix is registered with SFG_TRANSPOSE).

type Geno struct {
	g     *C.sfg_geno  // single device
	mg    *C.sfg_mgeno // multi-GPU engine: the matrix sharded by SNP block (exactly one of g / mg is set)
	Flags uint
	NRow  int
	NCol  int
}

// residentGeno is one registered matrix: its handle and stored shape.
type residentGeno struct {
	g          *C.sfg_geno
	mg         *C.sfg_mgeno
	nrow, ncol int
}

var (
	genoMu    sync.Mutex
	genoByKey = map[string]*Geno{}
	resident  []residentGeno // registered matrices, to find X when X^T is registered
)

const (
	FlagSquare    = uint(C.SFG_SQUARE)
	FlagTranspose = uint(C.SFG_TRANSPOSE)
)

// stagingRows is the number of matrix rows one chunk of the row-streamed registration carries: the staging buffer holds
// stagingRows * ncol bytes of page-locked memory (sfg_pinned_alloc), at most stagingBytes - nothing in this package scales with nrow * ncol.
const stagingBytes = 128 << 20

func stagingRows(ncol int) int {
	n := stagingBytes / ncol
	if n < 1 {
		n = 1
	}
	return n
}

// streamRows drives `visit(row0, nrows, chunk)` over the rows that next() yields (GenoFileStream.NextRow, filestream.go:414-426: one row per call, filters
// applied), nrows rows of ncol bytes at a time in a pinned staging buffer.  visit returns false to stop early.
func (h *Ctx) streamRows(nrow, ncol int, next func() []int8, visit func(row0, nrows int, chunk unsafe.Pointer) bool) {
	per := stagingRows(ncol)
	var buf unsafe.Pointer
	h.check(C.sfg_pinned_alloc(h.p, &buf, C.size_t(per*ncol)), "pinned_alloc")
	defer C.sfg_pinned_free(h.p, buf)
	stage := unsafe.Slice((*int8)(buf), per*ncol)
	for row0 := 0; row0 < nrow; row0 += per {
		n := per
		if nrow-row0 < n {
			n = nrow - row0
		}
		for r := 0; r < n; r++ {
			copy(stage[r*ncol:(r+1)*ncol], next())
		}
		if !visit(row0, n, buf) {
			return
		}
	}
}

// RegisterGeno makes the matrix behind a cache prefix resident, reading it the way the reference does - one row at a time (MatMult4StreamPreprocess,
// matmult.go:914-1041: gfs.NextRow per row) - so that no host allocation scales with nrow * ncol.  next() yields the rows in order (filters applied, missing = -1
// kept: the device zeroes negatives before sums and products, matmult.go:1292-1300); reset() rewinds the stream (GenoFileStream.Reset, filestream.go:362-376).
// If the TRANSPOSE of the arriving matrix is already resident (pca.go:112-113 registers X, then X^T from its own file) it is recognised on the device: every
// chunk of arriving rows is compared, entry by entry, with the resident matrix read as its transpose (sfg_geno_compare_rows) - exact, not a hash, and X^T is never
// held anywhere.  The prefix then becomes a SFG_TRANSPOSE view of the one int8 copy.  A shape match with different content (a second, unrelated n x n matrix) falls
// through to a second pass over the stream that uploads it.
func (h *Ctx) RegisterGeno(prefix string, nrow, ncol int, next func() []int8, reset func()) *Geno {
	genoMu.Lock()
	defer genoMu.Unlock()
	if g, ok := genoByKey[prefix]; ok {
		return g
	}
	for _, r := range resident {
		if r.nrow != ncol || r.ncol != nrow {
			continue
		}
		var ndiff C.uint64_t
		reset()
		h.streamRows(nrow, ncol, next, func(row0, n int, chunk unsafe.Pointer) bool {
			if r.mg != nil {
				h.mcheck(C.sfg_mgpu_geno_compare_rows(h.mg, r.mg, C.SFG_TRANSPOSE, C.size_t(row0), C.size_t(n), (*C.int8_t)(chunk), C.size_t(ncol), &ndiff), "mgpu_geno_compare_rows")
			} else {
				h.check(C.sfg_geno_compare_rows(h.p, r.g, C.SFG_TRANSPOSE, C.size_t(row0), C.size_t(n), (*C.int8_t)(chunk), C.size_t(ncol), &ndiff), "geno_compare_rows")
			}
			return ndiff == 0 // the first differing chunk settles it
		})
		if ndiff == 0 {
			e := &Geno{r.g, r.mg, FlagTranspose, nrow, ncol}
			genoByKey[prefix] = e
			return e
		}
	}
	var g *C.sfg_geno
	var m *C.sfg_mgeno
	if h.mg != nil {
		// the STORED orientation is the one whose columns are sharded over the GPUs: pca.go:112-113 registers X (individuals x SNPs) first, so SNP blocks
		// are the shards, Q * X is output-sharded and Q' * X^T contraction-sharded, as SURVEY 8e lays out
		h.mcheck(C.sfg_mgpu_geno_create(h.mg, C.size_t(nrow), C.size_t(ncol), &m), "mgpu_geno_create")
	} else {
		h.check(C.sfg_geno_create(h.p, C.size_t(nrow), C.size_t(ncol), &g), "geno_create")
	}
	reset()
	h.streamRows(nrow, ncol, next, func(row0, n int, chunk unsafe.Pointer) bool {
		if m != nil {
			h.mcheck(C.sfg_mgpu_geno_write_rows(h.mg, m, C.size_t(row0), C.size_t(n), (*C.int8_t)(chunk), C.size_t(ncol)), "mgpu_geno_write_rows")
		} else {
			h.check(C.sfg_geno_write_rows(h.p, g, C.size_t(row0), C.size_t(n), (*C.int8_t)(chunk), C.size_t(ncol)), "geno_write_rows")
		}
		return true
	})
	resident = append(resident, residentGeno{g, m, nrow, ncol})
	e := &Geno{g, m, 0, nrow, ncol}
	genoByKey[prefix] = e
	return e
}

// LookupGeno returns the resident matrix of a prefix, or nil (then MatMult4StreamCompute falls back to the DiagCache files on disk).
func LookupGeno(prefix string) *Geno {
	genoMu.Lock()
	defer genoMu.Unlock()
	return genoByKey[prefix]
}

// ----------------------------------------------------------------------------------------------------------------
// The calls the sibling files make.

// MatmulStream = MatMult4Stream on host buffers (include/sfgwas_hip.h: sfg_matmul_stream).
func (h *Ctx) MatmulStream(aFlat []uint64, s, inLevel, maxLevel int, geno []int8, nrow, ncol int, square bool, sum, sqsum []float64) []uint64 {
	mct := (ncol-1)/(h.N/2) + 1
	out := make([]uint64, s*mct*2*maxLevel*h.N)
	flags := C.uint(0)
	if square {
		flags |= C.SFG_SQUARE
	}
	var ps, pq *C.double
	if sum != nil {
		ps, pq = (*C.double)(unsafe.Pointer(&sum[0])), (*C.double)(unsafe.Pointer(&sqsum[0]))
	}
	h.check(C.sfg_matmul_stream(h.p, (*C.uint64_t)(unsafe.Pointer(&aFlat[0])), C.int(s), C.int(inLevel), C.int(maxLevel),
		(*C.int8_t)(unsafe.Pointer(&geno[0])), C.size_t(nrow), C.size_t(ncol), C.size_t(ncol), flags,
		(*C.uint64_t)(unsafe.Pointer(&out[0])), ps, pq), "matmul_stream")
	return out
}

// MatmulResident = MatMult4StreamCompute on a resident matrix; returns [s][m_ct][2][maxLevel][N] words.
func (h *Ctx) MatmulResident(aFlat []uint64, s, inLevel, maxLevel int, g *Geno) []uint64 {
	lcol := g.NCol
	if g.mg != nil { // every GPU of the node: sfg_mgpu_matmul shards, exchanges (Q' * X^T) and gathers
		mctM := (lcol-1)/(h.N/2) + 1
		out := make([]uint64, s*mctM*2*maxLevel*h.N)
		Default.mcheck(C.sfg_mgpu_matmul(Default.mg, (*C.uint64_t)(unsafe.Pointer(&aFlat[0])), C.int(s), C.int(inLevel), C.int(maxLevel), g.mg, C.uint(g.Flags),
			(*C.uint64_t)(unsafe.Pointer(&out[0]))), "mgpu_matmul")
		return out
	}
	dA := h.Upload(aFlat)
	defer dA.Free()
	mct := (lcol-1)/(h.N/2) + 1
	dOut := h.Alloc(8 * s * mct * 2 * maxLevel * h.N)
	defer dOut.Free()
	h.check(C.sfg_matmul_resident_dev(h.p, dA.U64(), C.int(s), C.int(inLevel), C.int(maxLevel), g.g, C.uint(g.Flags), dOut.U64()), "matmul_resident")
	return dOut.Download()
}

// AssocStreamPgen = the per-batch loop of GenoBlockMult's pgen branch (gwas/assoc.go:371-416) for one chromosome file, in one call: per batch of batchSnps kept
// variants FilterMatrixFilePgen + NewGenoFileStream + MatMult4Stream(cps, mat, X, 5, false, square, nproc), the batches' outputs in ConcatCipherMatrix order.
// sampleKeep: one byte per sample of the file (the --keep list of SampleKeepFile), snpFilt: one byte per variant of the file.  The baby-step rotations of `mat`
// are made once for all batches and kept on the device as the int8 MAC's rot tiles (include/sfgwas_hip.h: sfg_assoc_stream_pgen).  Returns [s][nct][2][maxLevel][N]
// words with nct = sum over batches of ceil(kept / slots).
func (h *Ctx) AssocStreamPgen(pgenPath string, sampleKeep, snpFilt []byte, batchSnps int, aFlat []uint64, s, inLevel, maxLevel int, square bool, capacity int) ([]uint64, int) {
	cp := C.CString(pgenPath)
	defer C.free(unsafe.Pointer(cp))
	flags := C.uint(0)
	if square {
		flags |= C.SFG_SQUARE
	}
	var pr, pc *C.uint8_t
	if sampleKeep != nil {
		pr = (*C.uint8_t)(unsafe.Pointer(&sampleKeep[0]))
	}
	if snpFilt != nil {
		pc = (*C.uint8_t)(unsafe.Pointer(&snpFilt[0]))
	}
	var nct C.size_t
	ctw := 2 * maxLevel * h.N
	compact := func(all []uint64) []uint64 { // [s][capacity][ctw] -> [s][nct][ctw]
		out := make([]uint64, s*int(nct)*ctw)
		for i := 0; i < s; i++ {
			copy(out[i*int(nct)*ctw:(i+1)*int(nct)*ctw], all[i*capacity*ctw:(i*capacity+int(nct))*ctw])
		}
		return out
	}
	if h.mg != nil { // batch k on GPU k % G (assoc.go:360-408 hands the batches to assoc_num_blocks_parallel workers the same way)
		nbr := len(aFlat) / (s * 2 * (inLevel + 1) * h.N)
		kept := nbr * (h.N / 2) // A's grid covers the kept samples; the library takes ceil(kept / slots) block rows from this
		all := make([]uint64, s*capacity*ctw)
		h.mcheck(C.sfg_mgpu_assoc_stream_pgen(h.mg, cp, pr, pc, C.size_t(kept), C.size_t(batchSnps), (*C.uint64_t)(unsafe.Pointer(&aFlat[0])), C.int(s), C.int(inLevel),
			C.int(maxLevel), flags, (*C.uint64_t)(unsafe.Pointer(&all[0])), C.size_t(capacity), &nct, nil, nil), "mgpu_assoc_stream_pgen")
		return compact(all), int(nct)
	}
	dA := h.Upload(aFlat)
	defer dA.Free()
	dOut := h.Alloc(8 * s * capacity * 2 * maxLevel * h.N)
	defer dOut.Free()
	h.check(C.sfg_assoc_stream_pgen(h.p, cp, pr, pc, C.size_t(batchSnps), dA.U64(), C.int(s), C.int(inLevel), C.int(maxLevel), flags,
		dOut.U64(), C.size_t(capacity), &nct, nil, nil), "assoc_stream_pgen") // computeSquaredSum = false on this branch (assoc.go:395)
	return compact(dOut.Download()), int(nct) // [s][capacity][2][maxLevel][N]: the first nct ciphertexts of every row
}

// MatmulFromCache = MatMult4StreamCompute on DiagCache files a CPU party wrote (gwas/filestream.go:19-282).
func (h *Ctx) MatmulFromCache(aFlat []uint64, s, inLevel, maxLevel int, prefix string, nbr int) []uint64 {
	cp := C.CString(prefix)
	defer C.free(unsafe.Pointer(cp))
	var hdr [6]C.uint64_t
	h.check(C.sfg_diagcache_header(h.p, cp, 0, &hdr[0]), "diagcache_header")
	mct := int(hdr[0])
	dA := h.Upload(aFlat)
	defer dA.Free()
	dOut := h.Alloc(8 * s * mct * 2 * maxLevel * h.N)
	defer dOut.Free()
	h.check(C.sfg_matmul_from_cache(h.p, dA.U64(), C.int(s), C.int(inLevel), C.int(maxLevel), cp, C.int(nbr), dOut.U64()), "matmul_from_cache")
	return dOut.Download()
}

// RotateRight: a batch of ciphertexts at one level, ct j rotated right by nrot[j] (crypto/basics.go:201-224 semantics).
func (h *Ctx) RotateRight(flat []uint64, nct, level int, nrot []int) []uint64 {
	dIn := h.Upload(flat)
	defer dIn.Free()
	dOut := h.Alloc(8 * len(flat))
	defer dOut.Free()
	cn := make([]C.int, nct)
	for i := range cn {
		cn[i] = C.int(nrot[i])
	}
	h.check(C.sfg_rotate_right_dev(h.p, dIn.U64(), dOut.U64(), C.int(nct), C.int(level), &cn[0]), "rotate_right")
	return dOut.Download()
}

// Binary element-wise ops on equal-level batches: "add", "sub", "mulrelin".
func (h *Ctx) Binary(op string, a, b []uint64, nct, level int) []uint64 {
	dA, dB := h.Upload(a), h.Upload(b)
	defer dA.Free()
	defer dB.Free()
	dOut := h.Alloc(8 * nct * h.CtWords(level))
	defer dOut.Free()
	switch op {
	case "add":
		h.check(C.sfg_ct_add_dev(h.p, dA.U64(), dB.U64(), dOut.U64(), C.int(nct), C.int(level)), op)
	case "sub":
		h.check(C.sfg_ct_sub_dev(h.p, dA.U64(), dB.U64(), dOut.U64(), C.int(nct), C.int(level)), op)
	case "mulrelin":
		h.check(C.sfg_ct_mulrelin_dev(h.p, dA.U64(), dB.U64(), dOut.U64(), C.int(nct), C.int(level)), op)
	default:
		panic("sfgwas-hip: unknown op " + op)
	}
	return dOut.Download()
}

// Rescale divides a batch at `level` by q_level (lattigo DivRoundByLastModulus); output at level-1.
func (h *Ctx) Rescale(in []uint64, nct, level int) []uint64 {
	dIn := h.Upload(in)
	defer dIn.Free()
	dOut := h.Alloc(8 * nct * h.CtWords(level-1))
	defer dOut.Free()
	h.check(C.sfg_ct_rescale_dev(h.p, dIn.U64(), dOut.U64(), C.int(nct), C.int(level)), "rescale")
	return dOut.Download()
}

// InnerSumAll: sum of nct ciphertexts, then the 13 rotate-and-add steps (crypto/basics.go:278-292); one ciphertext out.
func (h *Ctx) InnerSumAll(in []uint64, nct, level int) []uint64 {
	dIn := h.Upload(in)
	defer dIn.Free()
	dOut := h.Alloc(8 * h.CtWords(level))
	defer dOut.Free()
	h.check(C.sfg_ct_innersum_dev(h.p, dIn.U64(), C.int(nct), C.int(level), dOut.U64()), "innersum")
	return dOut.Download()
}

// MulPlain: ciphertext j times NTT-domain plaintext j (eval.MulRelinNew(plaintext, ct) behind crypto.CPMult / Mask, crypto/basics.go:110-172,429-470); one plaintext
// for all when npt == 1.  Level unchanged, the caller multiplies the scales.
func (h *Ctx) MulPlain(cts, pts []uint64, nct, npt, level int) []uint64 {
	dC, dP := h.Upload(cts), h.Upload(pts)
	defer dC.Free()
	defer dP.Free()
	dOut := h.Alloc(8 * nct * h.CtWords(level))
	defer dOut.Free()
	stride := 0
	if npt > 1 {
		stride = (level + 1) * h.N
	}
	h.check(C.sfg_ct_mul_plain_dev(h.p, dC.U64(), dP.U64(), C.size_t(stride), dOut.U64(), C.int(nct), C.int(level)), "mul_plain")
	return dOut.Download()
}

// ----------------------------------------------------------------------------------------------------------------
// Resident ciphertext vectors: the operands of CMult -> MatMult4StreamCompute -> InnerProd (gwas/matmult.go:27-77) stay in HBM between the calls instead of
// crossing PCIe four times.  A DevVec holds Len ciphertexts of ONE level and scale, [Len][2][Level+1][N].

type DevVec struct {
	Buf   *DevBuf
	Len   int
	Level int
	Scale float64
}

func (d *DevVec) Free() { d.Buf.Free() }

// UploadVec: the vector's ciphertexts at their common (minimum) level; their scales must agree (they do for every CipherVector the hot path builds:
// fresh encryptions and bootstrap outputs).
func (h *Ctx) UploadVec(v []*ckks.Ciphertext) *DevVec {
	level, scale := v[0].Level(), v[0].Scale()
	for _, c := range v {
		if c.Level() < level {
			level = c.Level()
		}
		if c.Scale() != scale {
			panic("sfgwas-hip: UploadVec: ciphertexts of one vector at different scales")
		}
	}
	return &DevVec{h.Upload(h.FlattenVec(v, len(v), level)), len(v), level, scale}
}

// DownloadVec: fresh ckks.Ciphertexts.
func (h *Ctx) DownloadVec(d *DevVec) []*ckks.Ciphertext {
	return h.VecFromFlat(d.Buf.Download(), d.Len, d.Level, d.Scale)
}

// dropTo returns a view of d at `level` (a new buffer when rows have to go: DropLevel keeps the first level + 1 rows of either polynomial).
func (h *Ctx) dropTo(d *DevVec, level int) (*DevVec, bool) {
	if d.Level == level {
		return d, false
	}
	out := h.Alloc(8 * d.Len * h.CtWords(level))
	h.check(C.sfg_ct_drop_level_dev(h.p, d.Buf.U64(), out.U64(), C.int(d.Len), C.int(d.Level), C.int(level)), "drop_level")
	return &DevVec{out, d.Len, level, d.Scale}, true
}

// CMultDev = crypto.CMult (crypto/basics.go:386-427) on resident operands: MulRelinNew + Rescale(minScale) per ciphertext, b broadcast when b.Len == 1.
// Every entry of a DevVec shares level and scale, so the whole vector is one (level, scale) group.
func (h *Ctx) CMultDev(a, b *DevVec, minScale float64) *DevVec {
	level := a.Level
	if b.Level < level {
		level = b.Level
	}
	n := a.Len
	if b.Len > n {
		n = b.Len
	}
	x, fx := h.dropTo(a, level)
	y, fy := h.dropTo(b, level)
	w := h.CtWords(level)
	rep := func(v *DevVec) *DevBuf { // length-1 broadcasting: n copies on the device
		if v.Len == n {
			return v.Buf
		}
		r := h.Alloc(8 * n * w)
		for i := 0; i < n; i++ {
			h.check(C.sfg_memcpy_d2d(h.p, unsafe.Pointer(uintptr(r.p)+uintptr(8*i*w)), v.Buf.p, C.size_t(8*w)), "d2d")
		}
		return r
	}
	bx, by := rep(x), rep(y)
	out := h.Alloc(8 * n * w)
	h.check(C.sfg_ct_mulrelin_dev(h.p, bx.U64(), by.U64(), out.U64(), C.int(n), C.int(level)), "mulrelin")
	if bx != x.Buf {
		bx.Free()
	}
	if by != y.Buf {
		by.Free()
	}
	if fx {
		x.Free()
	}
	if fy {
		y.Free()
	}
	scale := a.Scale * b.Scale
	qi := h.Params.Qi()
	for level > 0 && scale/float64(qi[level]) >= minScale/2 { // ckks.Evaluator.Rescale's loop (threshold scale)
		nxt := h.Alloc(8 * n * h.CtWords(level-1))
		h.check(C.sfg_ct_rescale_dev(h.p, out.U64(), nxt.U64(), C.int(n), C.int(level)), "rescale")
		out.Free()
		out = nxt
		scale /= float64(qi[level])
		level--
	}
	return &DevVec{out, n, level, scale}
}

// InnerSumAllDev = crypto.InnerSumAll (crypto/basics.go:278-292) on a resident vector: one ciphertext, the total in every slot.
func (h *Ctx) InnerSumAllDev(a *DevVec) *DevVec {
	out := h.Alloc(8 * h.CtWords(a.Level))
	h.check(C.sfg_ct_innersum_dev(h.p, a.Buf.U64(), C.int(a.Len), C.int(a.Level), out.U64()), "innersum")
	return &DevVec{out, 1, a.Level, a.Scale}
}

// MatmulResidentRows = MatMult4StreamCompute with the rows of A resident (each a DevVec of nbr ciphertexts at one common level): on a single device the input grid is
// assembled by device copies; with the multi-GPU engine the rows are downloaded once and the engine shards them.  Returns [s][m_ct][2][maxLevel][N] words.
func (h *Ctx) MatmulResidentRows(rows []*DevVec, maxLevel int, g *Geno) []uint64 {
	s, nbr, level := len(rows), rows[0].Len, rows[0].Level
	for _, r := range rows {
		if r.Len != nbr || r.Level != level {
			panic("sfgwas-hip: MatmulResidentRows: rows of different shape")
		}
	}
	w := h.CtWords(level)
	if g.mg != nil {
		flat := make([]uint64, s*nbr*w)
		for i, r := range rows {
			copy(flat[i*nbr*w:], r.Buf.Download())
		}
		return h.MatmulResident(flat, s, level, maxLevel, g)
	}
	grid := h.Alloc(8 * s * nbr * w)
	defer grid.Free()
	for i, r := range rows {
		h.check(C.sfg_memcpy_d2d(h.p, unsafe.Pointer(uintptr(grid.p)+uintptr(8*i*nbr*w)), r.Buf.p, C.size_t(8*nbr*w)), "d2d")
	}
	mct := (g.NCol-1)/(h.N/2) + 1
	dOut := h.Alloc(8 * s * mct * 2 * maxLevel * h.N)
	defer dOut.Free()
	h.check(C.sfg_matmul_resident_dev(h.p, grid.U64(), C.int(s), C.int(level), C.int(maxLevel), g.g, C.uint(g.Flags), dOut.U64()), "matmul_resident")
	return dOut.Download()
}

// BeaverElem / BeaverMatmul: local Beaver products over a prime field of `limbs` 64-bit little-endian limbs (mpc/beavermult.go:94-147).
func (h *Ctx) BeaverElem(pid, limbs int, modulus, ar, am, br, bm []uint64, n int) []uint64 {
	out := make([]uint64, n*limbs)
	p := func(s []uint64) *C.uint64_t {
		if len(s) == 0 {
			return nil
		}
		return (*C.uint64_t)(unsafe.Pointer(&s[0]))
	}
	h.check(C.sfg_beaver_elem(h.p, C.int(pid), C.int(limbs), p(modulus), p(ar), p(am), p(br), p(bm), p(out), C.size_t(n)), "beaver_elem")
	return out
}
func (h *Ctx) BeaverMatmul(pid, limbs int, modulus, ar, am, br, bm []uint64, m, k, n int) []uint64 {
	out := make([]uint64, m*n*limbs)
	p := func(s []uint64) *C.uint64_t {
		if len(s) == 0 {
			return nil
		}
		return (*C.uint64_t)(unsafe.Pointer(&s[0]))
	}
	h.check(C.sfg_beaver_matmul(h.p, C.int(pid), C.int(limbs), p(modulus), p(ar), p(am), p(br), p(bm), p(out), C.int(m), C.int(k), C.int(n)), "beaver_matmul")
	return out
}
