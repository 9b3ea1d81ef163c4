//go:build hip

// The pgen branch of GenoBlockMult (gwas/assoc.go:371-416) on the GPU library, for builds with `-tags hip`.
// In assoc.go the block between `if isPgen {` (:340) and `matOut = crypto.ConcatCipherMatrix(outMult)` (:416) becomes
//
//	matOut = genoBlockMultPgenHip(cryptoParams, pgenFile, sampleKeep, snpFilt, pgenBatchSize, mat, square, numCtx)
//	for c, kept := 0, 0; kept < nsnps; c++ { ... }   // filtOut as before: batch k marks filtOut[outShift_k .. outShift_k + counter_k)
//
// (behind `//go:build hip`; the untagged original keeps the shell-outs to plink2 / plinkBedToBinary.py).  dosageSum / dosageSqSum stay zero on this branch in the
// reference too (MatMult4Stream is called with computeSquaredSum = false and its sums are dropped, assoc.go:394-398).
// NOT COMPILED in the sfgwas-hip repository (no Go toolchain there); see integration/go/hip/hip.go.
package gwas

import (
	"github.com/hhcho/sfgwas/crypto"
	"github.com/hhcho/sfgwas/hip"
)

// boolsToBytes: the library's filters are one byte per sample / variant, zero = drop
func boolsToBytes(f []bool) []byte {
	if f == nil {
		return nil
	}
	b := make([]byte, len(f))
	for i, v := range f {
		if v {
			b[i] = 1
		}
	}
	return b
}

// genoBlockMultPgenHip multiplies `mat` (s rows of ceil(numInd / slots) ciphertexts) with every batch of pgenBatchSize kept variants of one chromosome's .pgen:
// what the dispatcher loop of assoc.go:371-412 does with FilterMatrixFilePgen + NewGenoFileStream + MatMult4Stream per batch and ConcatCipherMatrix at the end.
// sampleKeep: one flag per sample of the file, true = in SampleKeepFile (the reference hands plink2 the file itself; the caller reads it once per run);
// snpFilt: gwasParams.snpFilt[shift : shift+blockSize], nil = keep all; numCtx: the output width GenoBlockMult computed (:307-316) - the capacity of the call.
// The batches run one after the other on one context (the reference's LocalAssocNumBlocksParallel goroutines exist to keep CPU cores busy during the shell-outs;
// here the file is read ahead while the GPU multiplies, and the rotations of `mat` are shared by all batches instead of being redone per batch).
func genoBlockMultPgenHip(cps *crypto.CryptoParams, pgenFile string, sampleKeep, snpFilt []bool, pgenBatchSize int,
	mat crypto.CipherMatrix, square bool, numCtx int) crypto.CipherMatrix {
	h := hip.Default
	s, inLevel, maxLevel := len(mat), mat[0][0].Level(), 5
	aFlat := h.FlattenCipherMatrix(asRows(mat), inLevel)
	flat, nct := h.AssocStreamPgen(pgenFile+".pgen", boolsToBytes(sampleKeep), boolsToBytes(snpFilt), pgenBatchSize, aFlat, s, inLevel, maxLevel, square, numCtx)
	if nct != numCtx {
		panic("genoBlockMultPgenHip: the library produced a different number of output ciphertexts than GenoBlockMult expects")
	}
	return finish(cps, h, flat, s, nct, maxLevel, mat[0][0].Scale()*cps.Params.Scale())
}
