//go:build hip

// Drop-in bodies of the ciphertext-vector wrappers of crypto/basics.go that sit between the hot products (rows C1-C4 of
// SURVEY.md §8).  The untagged originals move behind `//go:build !hip`.  Each function keeps the reference's signature and
// scale / level bookkeeping; only the ring arithmetic runs on the device.  NOT COMPILED in the sfgwas-hip repository.
//
// These bodies round-trip through host memory per call, which is the minimal drop-in.  The resident form (ciphertexts stay
// on the device between CMult -> product -> InnerProd -> Sub) is what sfgwas_amd/host/gwas.hpp implements in C++ and what a
// maintainer would grow this file into (INTEGRATION.md §4b').
package crypto

import (
	"github.com/hhcho/sfgwas/hip"
	"github.com/ldsec/lattigo/v2/ckks"
	"go.dedis.ch/onet/v3/log"
)

func minLevel(v CipherVector) int {
	l := v[0].Level()
	for _, c := range v {
		if c.Level() < l {
			l = c.Level()
		}
	}
	return l
}

// rescaleLoop is ckks.Evaluator.Rescale(ct, minScale, ct) on a flat batch: divide by the last modulus while
// scale / q_level >= minScale / 2 (lattigo v2 ckks/evaluator.go: the threshold scale) and a level is left.
func rescaleLoop(h *hip.Ctx, flat []uint64, n, level int, scale, minScale float64) ([]uint64, int, float64) {
	qi := h.Params.Qi()
	for level > 0 && scale/float64(qi[level]) >= minScale/2 {
		flat = h.Rescale(flat, n, level)
		scale /= float64(qi[level])
		level--
	}
	return flat, level, scale
}

// RotateRightWithEvaluator - crypto/basics.go:201-210 (eva is unused: the device evaluator holds the keys).
func RotateRightWithEvaluator(cryptoParams *CryptoParams, ct *ckks.Ciphertext, nrot int, eva ckks.Evaluator) *ckks.Ciphertext {
	return RotateRight(cryptoParams, ct, nrot)
}

// RotateRight - crypto/basics.go:212-224: nrot mod slots, 0 = copy; otherwise eval.RotateNew(ct, slots - nrot).
func RotateRight(cryptoParams *CryptoParams, ct *ckks.Ciphertext, nrot int) *ckks.Ciphertext {
	nrot = Mod(nrot, cryptoParams.GetSlots())
	if nrot == 0 {
		return ct.CopyNew().Ciphertext()
	}
	h := hip.Default
	level := ct.Level()
	flat := make([]uint64, h.CtWords(level))
	h.FlattenCt(ct, level, flat)
	out := h.RotateRight(flat, 1, level, []int{nrot})
	return h.CtFromFlat(out, level, ct.Scale())
}

// CMult - crypto/basics.go:386-427: element-wise MulRelinNew + Rescale(Params.Scale()) with length-1 broadcasting.
func CMult(cryptoParams *CryptoParams, X CipherVector, Y CipherVector) CipherVector {
	h := hip.Default
	n := Max(len(X), len(Y))
	level := minLevel(X)
	if l := minLevel(Y); l < level {
		level = l
	}
	fx, fy := h.FlattenVec(X, n, level), h.FlattenVec(Y, n, level)
	prod := h.Binary("mulrelin", fx, fy, n, level)
	scale := X[0].Scale() * Y[0].Scale()
	prod, level, scale = rescaleLoop(h, prod, n, level, scale, cryptoParams.Params.Scale())
	return CipherVector(h.VecFromFlat(prod, n, level, scale))
}

// CAdd / CSub - crypto/basics.go:568-590 (equal scales and levels, as at every call site on the hot path; lattigo's scale
// matching for unequal scales is implemented in gwas.hpp's CAddSubDev and is PARITY UNPINNED - keep the CPU path for those).
func CAdd(cryptoParams *CryptoParams, X CipherVector, Y CipherVector) CipherVector {
	return addSub(cryptoParams, X, Y, "add")
}
func CSub(cryptoParams *CryptoParams, X CipherVector, Y CipherVector) CipherVector {
	return addSub(cryptoParams, X, Y, "sub")
}
func addSub(cryptoParams *CryptoParams, X, Y CipherVector, op string) CipherVector {
	h := hip.Default
	n := len(X)
	for i := range X {
		if X[i].Scale() != Y[i].Scale() {
			log.Fatal("sfgwas-hip: CAdd/CSub with unequal scales: use the CPU build for this call site")
		}
	}
	level := minLevel(X)
	if l := minLevel(Y); l < level {
		level = l
	}
	out := h.Binary(op, h.FlattenVec(X, n, level), h.FlattenVec(Y, n, level), n, level)
	return CipherVector(h.VecFromFlat(out, n, level, X[0].Scale()))
}

// InnerSumAll - crypto/basics.go:278-292: sum of the vector's ciphertexts, then 13 rotate-by-2^k-and-add steps.
func InnerSumAll(cryptoParams *CryptoParams, X CipherVector) *ckks.Ciphertext {
	h := hip.Default
	level := minLevel(X)
	out := h.InnerSumAll(h.FlattenVec(X, len(X), level), len(X), level)
	return h.CtFromFlat(out, level, X[0].Scale())
}

// CRescale - crypto/basics.go:707-719 (in place in the reference: the returned vector replaces X's entries).
func CRescale(cryptoParams *CryptoParams, X CipherVector) CipherVector {
	h := hip.Default
	for i := range X {
		level := X[i].Level()
		flat := make([]uint64, h.CtWords(level))
		h.FlattenCt(X[i], level, flat)
		out, l2, s2 := rescaleLoop(h, flat, 1, level, X[i].Scale(), cryptoParams.Params.Scale())
		X[i] = h.CtFromFlat(out, l2, s2)
	}
	return X
}
