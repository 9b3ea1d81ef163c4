//go:build hip

// Drop-in bodies of the three streaming products of gwas/matmult.go for builds with `-tags hip`.
// The untagged originals keep their names behind `//go:build !hip` (move MatMult4Stream, MatMult4StreamPreprocess and
// MatMult4StreamCompute of gwas/matmult.go:914-1505 into matmult_cpu.go with that tag; everything else in matmult.go is unchanged
// and shared).  NOT COMPILED in the sfgwas-hip repository (no Go toolchain there); see integration/go/hip/hip.go.
package gwas

import (
	"github.com/hhcho/sfgwas/crypto"
	"github.com/hhcho/sfgwas/hip"
	"github.com/hhcho/sfgwas/mpc"
	"github.com/ldsec/lattigo/v2/ckks"
)

// readAllRows drains a GenoFileStream into one row-major int8 matrix: filters applied by the stream (filestream.go:345-355,
// 419-423), values as stored (-1 = missing; the device zeroes negatives before sums and products as matmult.go:1292-1295 does).
// NOTE: construct the stream with replaceMissing = false or true - both give the same product; with true the -1s are already 0.
// Used by MatMult4Stream only, whose matrix is one association batch (assoc.go:371-424: 8192 SNPs of a chromosome file, or one block);
// the PCA matrices go through MatMult4StreamPreprocess below, which never holds them.
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
	// one row at a time, as the reference reads it (matmult.go:942-950: gfs.NextRow() per row of a block): hip.RegisterGeno fills a pinned staging buffer of
	// at most 128 MB from this callback and hands the chunks to the device - nothing here scales with nrow * ncol
	nrow, ncol := int(gfs.NumRowsToKeep()), int(gfs.NumColsToKeep())
	hip.Default.RegisterGeno(cacheFilePrefix, nrow, ncol, gfs.NextRow, gfs.Reset)
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

// QXLazyNormStream - gwas/matmult.go:27-77 (move the original behind `//go:build !hip` with the three products): Q * S * (X - m 1^T) as (Q S) X - ((Q S) m) 1^T.
// Q S is made ON the device (crypto.CMult, basics.go:386-427) and stays there for its two consumers - the product and the inner products with XMean - instead of
// being downloaded by CMult, uploaded by MatMult4StreamCompute and uploaded again by every InnerProd.  The network step (BootstrapMatAll) and the final Sub /
// MaskTrunc (lattigo's own scale matching) are the reference's lines, unchanged.
func QXLazyNormStream(cps *crypto.CryptoParams, mpcObj *mpc.MPC, Q crypto.CipherMatrix, Xcachefile string, XMean, XStdInv crypto.CipherVector, numInd int) (out crypto.CipherMatrix) {
	if mpcObj.GetPid() == 0 {
		return
	}
	h := hip.Default
	slots := cps.GetSlots()
	g := hip.LookupGeno(Xcachefile)
	if g == nil { // DiagCache files of a CPU run: the host-pointer path
		QS := make(crypto.CipherMatrix, len(Q))
		for i := range Q {
			QS[i] = crypto.CMult(cps, Q[i], XStdInv)
		}
		return qxLazyNormTail(cps, mpcObj, MatMult4StreamCompute(cps, QS, 5, Xcachefile), func(i int) *ckks.Ciphertext { return crypto.InnerProd(cps, QS[i], XMean) }, slots, numInd)
	}
	dS, dM := h.UploadVec([]*ckks.Ciphertext(XStdInv)), h.UploadVec([]*ckks.Ciphertext(XMean))
	defer dS.Free()
	defer dM.Free()
	QS := make([]*hip.DevVec, len(Q))
	for i := range Q { // QS[i] = crypto.CMult(cps, Q[i], XStdInv)
		dq := h.UploadVec([]*ckks.Ciphertext(Q[i]))
		QS[i] = h.CMultDev(dq, dS, cps.Params.Scale())
		dq.Free()
	}
	defer func() {
		for _, v := range QS {
			v.Free()
		}
	}()
	flat := h.MatmulResidentRows(QS, 5, g) // out = MatMult4StreamCompute(cps, QS, 5, Xcachefile)
	mct := (g.NCol-1)/slots + 1
	prod := finish(cps, h, flat, len(Q), mct, 5, QS[0].Scale*cps.Params.Scale())
	return qxLazyNormTail(cps, mpcObj, prod, func(i int) *ckks.Ciphertext { // crypto.InnerProd(cps, QS[i], XMean) = InnerSumAll(CMult(QS[i], XMean)), basics.go:274-292
		p := h.CMultDev(QS[i], dM, cps.Params.Scale())
		t := h.InnerSumAllDev(p)
		ct := h.DownloadVec(t)[0]
		p.Free()
		t.Free()
		return ct
	}, slots, numInd)
}

// qxLazyNormTail: gwas/matmult.go:44-72 as written there - bootstrap, (Q S) m per row, Sub, MaskTrunc of the ragged tail.
func qxLazyNormTail(cps *crypto.CryptoParams, mpcObj *mpc.MPC, out crypto.CipherMatrix, innerProd func(int) *ckks.Ciphertext, slots, numInd int) crypto.CipherMatrix {
	out = mpcObj.Network.BootstrapMatAll(cps, out)
	for i := range out {
		QSm := innerProd(i)
		cps.WithEvaluator(func(eval ckks.Evaluator) error {
			for j := range out[i] {
				eval.Sub(out[i][j], QSm, out[i][j])
			}
			return nil
		})
		for j := range out[i] {
			N := slots
			if j == len(out[i])-1 {
				N = ((numInd - 1) % slots) + 1
			}
			out[i][j] = crypto.MaskTrunc(cps, out[i][j], N)
		}
	}
	return out
}
