//go:build hip

// Drop-in bodies of the ciphertext-vector wrappers of crypto/basics.go that sit between the hot products (rows C1-C4 of
// SURVEY.md §8).  The untagged originals move behind `//go:build !hip`.  Each function keeps the reference's signature and
// scale / level bookkeeping; only the ring arithmetic runs on the device.  NOT COMPILED in the sfgwas-hip repository.
//
// These bodies round-trip through host memory per call, which is the minimal drop-in.  The resident form (ciphertexts stay on the device between
// CMult -> product -> InnerProd) is hip.DevVec + gwas/matmult_hip.go's QXLazyNormStream; the C++ mirror sfgwas_amd/host/gwas.hpp carries it through Sub / MaskTrunc too.
package crypto

import (
	"github.com/hhcho/sfgwas/hip"
	"github.com/ldsec/lattigo/v2/ckks"
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

// pick returns v[i] with length-1 broadcasting (crypto/basics.go:386-427 treats a one-element operand as a scalar).
func pick(v CipherVector, i int) *ckks.Ciphertext {
	if len(v) == 1 {
		return v[0]
	}
	return v[i]
}

// opGroup: the entries of an element-wise vector op that share operand level and scales - ONE batched device call each.  The reference loops
// eval.MulRelinNew / Rescale / AddNew per ciphertext (crypto/basics.go:386-470, 568-590): every result keeps its OWN level (min of its two operands) and scale;
// dropping a whole vector to its minimum level would change the words of the higher entries (qrfact.go's Householder vectors carry their pivot ciphertext one
// level below the others).
type opGroup struct {
	level  int
	sx, sy float64
}

func groupPairs(n int, lx, ly func(int) int, sx, sy func(int) float64) ([]opGroup, map[opGroup][]int) {
	idx := map[opGroup][]int{}
	var order []opGroup
	for i := 0; i < n; i++ {
		l := lx(i)
		if ly(i) < l {
			l = ly(i)
		}
		k := opGroup{l, sx(i), sy(i)}
		if _, ok := idx[k]; !ok {
			order = append(order, k)
		}
		idx[k] = append(idx[k], i)
	}
	return order, idx
}

func flattenPicked(h *hip.Ctx, v CipherVector, ids []int, level int) []uint64 {
	w := h.CtWords(level)
	flat := make([]uint64, len(ids)*w)
	for j, i := range ids {
		h.FlattenCt(pick(v, i), level, flat[j*w:(j+1)*w])
	}
	return flat
}

// CMult - crypto/basics.go:386-427: element-wise MulRelinNew + Rescale(Params.Scale()) with length-1 broadcasting, per ciphertext level and scale.
func CMult(cryptoParams *CryptoParams, X CipherVector, Y CipherVector) CipherVector {
	h := hip.Default
	n := Max(len(X), len(Y))
	res := make(CipherVector, n)
	order, idx := groupPairs(n, func(i int) int { return pick(X, i).Level() }, func(i int) int { return pick(Y, i).Level() },
		func(i int) float64 { return pick(X, i).Scale() }, func(i int) float64 { return pick(Y, i).Scale() })
	for _, k := range order {
		ids := idx[k]
		prod := h.Binary("mulrelin", flattenPicked(h, X, ids, k.level), flattenPicked(h, Y, ids, k.level), len(ids), k.level)
		prod, level, scale := rescaleLoop(h, prod, len(ids), k.level, k.sx*k.sy, cryptoParams.Params.Scale())
		for j, ct := range h.VecFromFlat(prod, len(ids), level, scale) {
			res[ids[j]] = ct
		}
	}
	return res
}

// CPMult - crypto/basics.go:429-470: the same with NTT-domain plaintexts (eval.MulRelinNew(ct, plaintext) multiplies both polynomials by the plaintext).
func CPMult(cryptoParams *CryptoParams, X CipherVector, Y PlainVector) CipherVector {
	h := hip.Default
	n := Max(len(X), len(Y))
	res := make(CipherVector, n)
	py := func(i int) *ckks.Plaintext {
		if len(Y) == 1 {
			return Y[0]
		}
		return Y[i]
	}
	order, idx := groupPairs(n, func(i int) int { return pick(X, i).Level() }, func(i int) int { return py(i).Level() },
		func(i int) float64 { return pick(X, i).Scale() }, func(i int) float64 { return py(i).Scale() })
	for _, k := range order {
		ids := idx[k]
		nl := k.level + 1
		pts := make([]uint64, len(ids)*nl*h.N)
		for j, i := range ids {
			for l := 0; l < nl; l++ {
				copy(pts[(j*nl+l)*h.N:], py(i).Value()[0].Coeffs[l][:h.N])
			}
		}
		prod := h.MulPlain(flattenPicked(h, X, ids, k.level), pts, len(ids), len(ids), k.level)
		prod, level, scale := rescaleLoop(h, prod, len(ids), k.level, k.sx*k.sy, cryptoParams.Params.Scale())
		for j, ct := range h.VecFromFlat(prod, len(ids), level, scale) {
			res[ids[j]] = ct
		}
	}
	return res
}

// CAdd / CSub - crypto/basics.go:568-590: eval.AddNew / SubNew per ciphertext - result level = min of the pair.  Pairs at EQUAL scales only (every call site on
// the PCA path); lattigo's scale matching for unequal scales is restated in gwas.hpp's CellVec ops and is PARITY UNPINNED - such pairs stay on the CPU evaluator.
func CAdd(cryptoParams *CryptoParams, X CipherVector, Y CipherVector) CipherVector {
	return addSub(cryptoParams, X, Y, "add")
}
func CSub(cryptoParams *CryptoParams, X CipherVector, Y CipherVector) CipherVector {
	return addSub(cryptoParams, X, Y, "sub")
}
func addSub(cryptoParams *CryptoParams, X, Y CipherVector, op string) CipherVector {
	h := hip.Default
	n := len(X)
	res := make(CipherVector, n)
	order, idx := groupPairs(n, func(i int) int { return X[i].Level() }, func(i int) int { return Y[i].Level() },
		func(i int) float64 { return X[i].Scale() }, func(i int) float64 { return Y[i].Scale() })
	for _, k := range order {
		ids := idx[k]
		if k.sx != k.sy { // lattigo rescales one operand by the integer scale ratio first: left to lattigo itself
			cryptoParams.WithEvaluator(func(eval ckks.Evaluator) error {
				for _, i := range ids {
					if op == "add" {
						res[i] = eval.AddNew(X[i], Y[i])
					} else {
						res[i] = eval.SubNew(X[i], Y[i])
					}
				}
				return nil
			})
			continue
		}
		out := h.Binary(op, flattenPicked(h, X, ids, k.level), flattenPicked(h, Y, ids, k.level), len(ids), k.level)
		for j, ct := range h.VecFromFlat(out, len(ids), k.level, k.sx) {
			res[ids[j]] = ct
		}
	}
	return res
}

// InnerSumAll - crypto/basics.go:278-292: sum of the vector's ciphertexts, then 13 rotate-by-2^k-and-add steps.
// The reference adds ciphertext by ciphertext with eval.Add (:283-288), i.e. with lattigo's own level and scale matching between operands.  Where every
// ciphertext of X has the same level and scale - every call site of the PCA path: matmult.go:96,137,175 sum the outputs of one CMult, pca.go:477 and
// qrfact.go:364 pass a single ciphertext - the sum runs on the device with the rotations.  A vector of mixed levels or scales (assoc.go:813,819 may see
// one after a Rebalance, the only call sites that can) is summed by the reference's own lines on the host, so that whatever lattigo's Add does about the
// mismatch is what happens here too; the device then takes the one summed ciphertext through the 13 rotation steps.
func InnerSumAll(cryptoParams *CryptoParams, X CipherVector) *ckks.Ciphertext {
	h := hip.Default
	uniform := true
	for i := 1; i < len(X); i++ {
		if X[i].Level() != X[0].Level() || X[i].Scale() != X[0].Scale() {
			uniform = false
			break
		}
	}
	if uniform {
		level := X[0].Level()
		out := h.InnerSumAll(h.FlattenVec(X, len(X), level), len(X), level)
		return h.CtFromFlat(out, level, X[0].Scale())
	}
	vecsum := X[0].CopyNew().Ciphertext() // basics.go:280-288, verbatim
	for i := 1; i < len(X); i++ {
		cryptoParams.WithEvaluator(func(eval ckks.Evaluator) error {
			eval.Add(X[i], vecsum, vecsum)
			return nil
		})
	}
	level := vecsum.Level()
	one := CipherVector{vecsum}
	out := h.InnerSumAll(h.FlattenVec(one, 1, level), 1, level)
	return h.CtFromFlat(out, level, vecsum.Scale())
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
