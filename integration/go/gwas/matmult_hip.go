//go:build hip

// Drop-in bodies of the three streaming products of gwas/matmult.go for builds with `-tags hip`.
// The untagged originals keep their names behind `//go:build !hip` (move MatMult4Stream, MatMult4StreamPreprocess and
// MatMult4StreamCompute of gwas/matmult.go:914-1505 into matmult_cpu.go with that tag; everything else in matmult.go is unchanged
// and shared).  NOT COMPILED in the sfgwas-hip repository (no Go toolchain there); see integration/go/hip/hip.go.
package gwas

import (
	"github.com/hhcho/sfgwas/crypto"
	"github.com/hhcho/sfgwas/hip"
	"github.com/ldsec/lattigo/v2/ckks"
)

// readAllRows drains a GenoFileStream into one row-major int8 matrix: filters applied by the stream (filestream.go:345-355,
// 419-423), values as stored (-1 = missing; the device zeroes negatives before sums and products as matmult.go:1292-1295 does).
// NOTE: construct the stream with replaceMissing = false or true - both give the same product; with true the -1s are already 0.
func readAllRows(gfs *GenoFileStream) ([]int8, int, int) {
	gfs.Reset()
	nrow, ncol := int(gfs.NumRowsToKeep()), int(gfs.NumColsToKeep())
	geno := make([]int8, nrow*ncol)
	for r := 0; r < nrow; r++ {
		copy(geno[r*ncol:(r+1)*ncol], gfs.NextRow())
	}
	return geno, nrow, ncol
}

func toCipherMatrix(rows [][]*ckks.Ciphertext) crypto.CipherMatrix {
	out := make(crypto.CipherMatrix, len(rows))
	for i := range rows {
		out[i] = crypto.CipherVector(rows[i])
	}
	return out
}

func asRows(A crypto.CipherMatrix) [][]*ckks.Ciphertext {
	rows := make([][]*ckks.Ciphertext, len(A))
	for i := range A {
		rows[i] = []*ckks.Ciphertext(A[i])
	}
	return rows
}

// finish: the reference starts from crypto.CZeroMat - fresh encryptions of zero - and adds the aggregated giant steps onto it
// (matmult.go:1174,1225 / :1443,1494); the library returns the deterministic sum, level maxLevel-1, scale A.scale * Params.Scale (:1045,350).
func finish(cps *crypto.CryptoParams, h *hip.Ctx, flat []uint64, s, mct, maxLevel int, outScale float64) crypto.CipherMatrix {
	out := crypto.CZeroMat(cps, mct, s) // s rows of mct ciphertexts (basics.go:378-384: CZeroMat(cryptoParams, nrows=mct, ncols=s) -> [s][mct])
	cps.WithEvaluator(func(eval ckks.Evaluator) error {
		h.AddFlatInto(eval, asRows(out), flat, maxLevel-1, outScale)
		return nil
	})
	return out
}

// MatMult4Stream - gwas/matmult.go:1238-1505, association path: on-the-fly encode from int8 rows.
func MatMult4Stream(cps *crypto.CryptoParams, A crypto.CipherMatrix, gfs *GenoFileStream, maxLevel int,
	computeSquaredSum, square bool, nproc int) (crypto.CipherMatrix, []float64, []float64) {
	h := hip.Default.Fork() // GenoBlockMult runs assoc_num_blocks_parallel of these at once (assoc.go:360-408)
	defer h.Close()
	geno, nrow, ncol := readAllRows(gfs)
	s, inLevel := len(A), A[0][0].Level()
	aFlat := h.FlattenCipherMatrix(asRows(A), inLevel) // the library drops the inputs to maxLevel itself (matmult.go:1256-1259)
	var sum, sq []float64
	if computeSquaredSum {
		sum, sq = make([]float64, ncol), make([]float64, ncol)
	}
	flat := h.MatmulStream(aFlat, s, inLevel, maxLevel, geno, nrow, ncol, square, sum, sq)
	mct := (ncol-1)/cps.GetSlots() + 1
	return finish(cps, h, flat, s, mct, maxLevel, A[0][0].Scale()*cps.Params.Scale()), sum, sq
}

// MatMult4StreamPreprocess - gwas/matmult.go:914-1041.  The reference writes every encoded diagonal to <prefix>_<bi>.bin
// (96 bytes per genotype); here the int8 matrix becomes resident in HBM (1 byte per genotype) and the prefix is its key.
// pca.go:112-113 calls this once for X and once for X^T: the second call finds the first matrix and registers the transposed view.
func MatMult4StreamPreprocess(cps *crypto.CryptoParams, gfs *GenoFileStream, maxLevel int, cacheFilePrefix string) {
	if hip.LookupGeno(cacheFilePrefix) != nil {
		return // "skips existing files" (matmult.go:928-931)
	}
	geno, nrow, ncol := readAllRows(gfs)
	hip.Default.RegisterGeno(cacheFilePrefix, geno, nrow, ncol)
}

// MatMult4StreamCompute - gwas/matmult.go:1043-1236, PCA path.
func MatMult4StreamCompute(cps *crypto.CryptoParams, A crypto.CipherMatrix, maxLevel int, cacheFilePrefix string) crypto.CipherMatrix {
	h := hip.Default
	s, inLevel := len(A), A[0][0].Level()
	aFlat := h.FlattenCipherMatrix(asRows(A), inLevel)
	outScale := A[0][0].Scale() * cps.Params.Scale()
	if g := hip.LookupGeno(cacheFilePrefix); g != nil {
		flat := h.MatmulResident(aFlat, s, inLevel, maxLevel, g)
		mct := (g.NCol-1)/cps.GetSlots() + 1
		return finish(cps, h, flat, s, mct, maxLevel, outScale)
	}
	// no resident matrix under this prefix: the DiagCache files of a CPU run (filestream.go:19-282) are multiplied as they are
	nbr := len(A[0])
	flat := h.MatmulFromCache(aFlat, s, inLevel, maxLevel, cacheFilePrefix, nbr)
	mct := len(flat) / (s * 2 * maxLevel * cps.Params.N())
	return finish(cps, h, flat, s, mct, maxLevel, outScale)
}
