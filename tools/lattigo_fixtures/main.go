// lattigo_fixtures — dumps golden vectors through the REFERENCE's own lattigo call sites, so that the CPU oracle of the
// MI355X build (oracle/sfgwas_oracle.c) can be pinned bit-for-bit against the pinned fork
//   github.com/ldsec/lattigo/v2 => github.com/hcholab/lattigo/v2 v2.1.2-0.20230123224332-e8d68c24b94a   (reference go.mod:5,12)
//
// This program cannot be built in the build image (no Go toolchain, no module cache, no network).  Run it on any machine with
// Go >= 1.18 and network access (see README.md), then convert the dump with to_npz.py and commit tests/golden/lattigo_vectors.npz.
// Every call below is one the reference itself makes (file:line of the reference given), so it exists in the fork.
//
// Output: one container file of named little-endian arrays (format: see writeArr).
package main

import (
	"encoding/binary"
	"fmt"
	"math"
	"os"

	"github.com/ldsec/lattigo/v2/ckks"
	"github.com/ldsec/lattigo/v2/dckks"
	"github.com/ldsec/lattigo/v2/ring"
)

var out *os.File

// record: u32 name length, name, u32 dtype (0 = uint64, 1 = float64), u32 ndim, ndim x u64 dims, data (little endian)
func writeArr(name string, dtype uint32, dims []uint64, data []uint64) {
	binary.Write(out, binary.LittleEndian, uint32(len(name)))
	out.Write([]byte(name))
	binary.Write(out, binary.LittleEndian, dtype)
	binary.Write(out, binary.LittleEndian, uint32(len(dims)))
	binary.Write(out, binary.LittleEndian, dims)
	binary.Write(out, binary.LittleEndian, data)
}
func u64s(name string, v []uint64)   { writeArr(name, 0, []uint64{uint64(len(v))}, v) }
func f64s(name string, v []float64) {
	b := make([]uint64, len(v))
	for i := range v {
		b[i] = math.Float64bits(v[i])
	}
	writeArr(name, 1, []uint64{uint64(len(v))}, b)
}
func poly(name string, p *ring.Poly, rows int) { // Coeffs[0..rows) flattened [rows][N]
	n := len(p.Coeffs[0])
	flat := make([]uint64, 0, rows*n)
	for l := 0; l < rows; l++ {
		flat = append(flat, p.Coeffs[l]...)
	}
	writeArr(name, 0, []uint64{uint64(rows), uint64(n)}, flat)
}
func ct(name string, c *ckks.Ciphertext) { // [2][level+1][N], crypto.go:32-60 layout
	rows := int(c.Level()) + 1
	poly(name+".c0", c.Value()[0], rows)
	poly(name+".c1", c.Value()[1], rows)
	f64s(name+".scale", []float64{c.Scale()})
}

// deterministic test data (splitmix64, the generator shared with oracle/sfgwas_oracle.c orc_splitmix64)
func splitmix(state *uint64) uint64 {
	*state += 0x9E3779B97F4A7C15
	z := *state
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9
	z = (z ^ (z >> 27)) * 0x94D049BB133111EB
	return z ^ (z >> 31)
}
func genoVector(seed uint64, n int) []float64 { // an int8 genotype diagonal: values 0,1,2 (matmult.go:636-664 feeds these to EncodeNTT)
	v := make([]float64, n)
	for i := range v {
		v[i] = float64(splitmix(&seed) % 3)
	}
	return v
}
func toComplex(v []float64) []complex128 { // convertToComplex128WithRot(buf, 0), matmult.go:666-672
	c := make([]complex128, len(v))
	for i := range v {
		c[i] = complex(v[i], 0)
	}
	return c
}

func dumpParams(tag string, params *ckks.Parameters) {
	u64s(tag+".qi", params.Qi()) // matmult.go:328
	u64s(tag+".pi", params.Pi())
	u64s(tag+".logN_logSlots_maxLevel", []uint64{params.LogN(), params.LogSlots(), params.MaxLevel()})
	f64s(tag+".scale", []float64{params.Scale()})
}

func run(tag string, params *ckks.Parameters, level uint64, withKeys bool) {
	dumpParams(tag, params)
	slots := int(params.Slots())
	// --- ring.NTT of a fixed row per modulus (behind every lattigo op the hot path uses)
	ringQP, _ := ring.NewRing(params.N(), append(params.Qi(), params.Pi()...)) // crypto.go:164
	row := ringQP.NewPoly()
	st := uint64(0xA11CE)
	for l := range row.Coeffs {
		for j := range row.Coeffs[l] {
			row.Coeffs[l][j] = splitmix(&st) % ringQP.Modulus[l]
		}
	}
	poly(tag+".ntt.in", row, len(row.Coeffs))
	ringQP.NTT(row, row)
	poly(tag+".ntt.out", row, len(row.Coeffs))
	// --- EncodeNTT with the big-float encoder the reference uses (matmult.go:723,1019,1421; prec = mpc_field_size = 256, gwas.go:180,212)
	enc := ckks.NewEncoderBig(params, 256)
	for k, seed := range []uint64{1, 2, 3} {
		v := genoVector(seed, slots)
		pt := ckks.NewPlaintext(params, level, params.Scale()) // matmult.go:722
		enc.EncodeNTT(pt, toComplex(v), params.LogSlots())     // matmult.go:723
		f64s(fmt.Sprintf("%s.encode%d.values", tag, k), v)
		poly(fmt.Sprintf("%s.encode%d.pt", tag, k), pt.Value()[0], int(level)+1)
		if k == 0 { // DiagCache payload bytes of this plaintext (filestream.go:217): byte order of ring.WriteCoeffsTo
			buf := make([]byte, 8*int(params.N())*(int(level)+1)+16)
			n, _ := ring.WriteCoeffsTo(0, int(params.N()), int(level)+1, pt.Value()[0].Coeffs, buf)
			w := make([]uint64, n)
			for i := 0; i < n; i++ {
				w[i] = uint64(buf[i])
			}
			u64s(tag+".encode0.writecoeffs_bytes", w)
		}
	}
	if !withKeys {
		return
	}
	// --- keys exactly as crypto.go:159-210 makes them
	kgen := ckks.NewKeyGenerator(params)
	sk := kgen.GenSecretKey()
	pk := kgen.GenPublicKey(sk)
	rlk := kgen.GenRelinearizationKey(sk)
	rots := []int{1, 91, slots - 1}
	rotKs := kgen.GenRotationKeysForRotations(rots, false, sk) // crypto.go:208
	poly(tag+".sk", sk.Value, len(sk.Value.Coeffs))
	for _, k := range rots {
		galEl := params.GaloisElementForColumnRotationBy(k) // mhe.go:404
		swk := rotKs.Keys[galEl]                             // mhe.go:469
		u64s(fmt.Sprintf("%s.rot%d.galois", tag, k), []uint64{galEl})
		for i := range swk.Value { // [beta][2] polys over Q u P, NTT + Montgomery form as lattigo stores them
			poly(fmt.Sprintf("%s.rot%d.key.%d.0", tag, k, i), swk.Value[i][0], len(swk.Value[i][0].Coeffs))
			poly(fmt.Sprintf("%s.rot%d.key.%d.1", tag, k, i), swk.Value[i][1], len(swk.Value[i][1].Coeffs))
		}
	}
	for i := range rlk.Keys[0].Value {
		poly(fmt.Sprintf("%s.rlk.%d.0", tag, i), rlk.Keys[0].Value[i][0], len(rlk.Keys[0].Value[i][0].Coeffs))
		poly(fmt.Sprintf("%s.rlk.%d.1", tag, i), rlk.Keys[0].Value[i][1], len(rlk.Keys[0].Value[i][1].Coeffs))
	}
	// --- two fresh ciphertexts (crypto.go:325-340)
	encryptor := ckks.NewEncryptorFromPk(params, pk)
	eval := ckks.NewEvaluator(params, ckks.EvaluationKey{Rlk: rlk, Rtks: rotKs}) // matmult.go:1110
	mk := func(seed uint64) *ckks.Ciphertext {
		pt := ckks.NewPlaintext(params, params.MaxLevel(), params.Scale())
		enc.EncodeNTT(pt, toComplex(genoVector(seed, slots)), params.LogSlots())
		return encryptor.EncryptNew(pt)
	}
	a, b := mk(11), mk(12)
	a = eval.DropLevelNew(a, params.MaxLevel()-level) // basics.go:813 (matmult.go:1055 drops to maxLevel)
	b = eval.DropLevelNew(b, params.MaxLevel()-level)
	ct(tag+".a", a)
	ct(tag+".b", b)
	for _, k := range rots { // crypto.RotateRightWithEvaluator -> RotateNew(ct, slots - nrot), basics.go:205
		ct(fmt.Sprintf("%s.rotate_left_%d", tag, k), eval.RotateNew(a, k))
	}
	prod := eval.MulRelinNew(a, b) // basics.go:393
	ct(tag+".mulrelin", prod)
	eval.Rescale(prod, params.Scale(), prod) // basics.go:394
	ct(tag+".mulrelin_rescaled", prod)
	cm := eval.MultByConstNew(a, 1.0/8192.0) // basics.go:489
	ct(tag+".multbyconst_1_8192", cm)
	ac := eval.AddConstNew(a, 0.5) // basics.go:195
	ct(tag+".addconst_0p5", ac)
	sum := eval.AddNew(a, b)
	ct(tag+".add", sum)
	// --- collective bootstrap, local work, at the scales the reference uses (mpc/mhe.go:245-258 with one party): a product at scale Delta^2 is
	// refreshed to parameters.Scale().  GenShares draws its mask internally; what the dump pins is (i) Recode as a function of the Decrypt output
	// (scale ratio and truncation rule) and (ii) the relation between the two shares (h0 - sk*c1 vs -h1 - sk*crp carry mask and scaled mask).
	{
		refProtocol := dckks.NewRefreshProtocol(params)
		in := eval.MulRelinNew(a, b) // scale Delta^2, level `level`
		ct(tag+".refresh.in", in)
		levelStart := in.Level()
		refShare1, refShare2 := refProtocol.AllocateShares(levelStart) // mhe.go:246
		ringQ, _ := ring.NewRing(params.N(), params.Qi())
		crp := ringQ.NewPoly() // stands for crpGen.ReadNew(), mhe.go:248
		cst := uint64(0xC0FFEE)
		for l := range crp.Coeffs {
			for j := range crp.Coeffs[l] {
				crp.Coeffs[l][j] = splitmix(&cst) % ringQ.Modulus[l]
			}
		}
		poly(tag+".refresh.crp", crp, len(crp.Coeffs))
		refProtocol.GenShares(sk.Value, levelStart, 1, in, params.Scale(), crp, refShare1, refShare2) // mhe.go:251
		poly(tag+".refresh.h0", (*ring.Poly)(refShare1), int(levelStart)+1)
		poly(tag+".refresh.h1", (*ring.Poly)(refShare2), int(params.MaxLevel())+1)
		refProtocol.Decrypt(in, refShare1) // mhe.go:256 (one party: the aggregate is the share)
		poly(tag+".refresh.decrypted_c0", in.Value()[0], int(levelStart)+1)
		refProtocol.Recode(in, params.Scale()) // mhe.go:257
		poly(tag+".refresh.recoded_c0", in.Value()[0], int(params.MaxLevel())+1)
		f64s(tag+".refresh.recoded_scale", []float64{in.Scale()})
		refProtocol.Recrypt(in, crp, refShare2) // mhe.go:258
		ct(tag+".refresh.out", in)
	}
	// decrypted slots of the rotation (sanity, fp64): crypto.go:451-455
	dec := ckks.NewDecryptor(params, sk)
	vals := enc.Decode(dec.DecryptNew(eval.RotateNew(a, 1)), params.LogSlots())
	re := make([]float64, 8)
	for i := range re {
		re[i] = real(vals[i])
	}
	f64s(tag+".rotate_left_1.decoded_head", re)
}

func main() {
	var err error
	out, err = os.Create("lattigo_vectors.bin")
	if err != nil {
		panic(err)
	}
	defer out.Close()
	// (1) the preset of the shipped configuration (config/configGlobal.toml:8, gwas.go:169): moduli, NTT, EncodeNTT at maxLevel = 5
	run("pn14", ckks.DefaultParams[ckks.PN14QP438], 5, false)
	// (2) a small ring with the same SHAPE of modulus chain (46-bit q0, 35-bit q1..q5, two 43-bit special primes -> alpha = 2,
	//     beta = 3 at level 5): small enough to commit complete switching keys, and it exercises the multi-prime basis extension
	small := ckks.NewParametersFromLogModuli(10, 9, 1<<34, ckks.LogModuli{LogQi: []uint64{46, 35, 35, 35, 35, 35}, LogPi: []uint64{43, 43}}, 3.2) // crypto.go:82 (commented call shape)
	run("small", small, 5, true)
	fmt.Println("wrote lattigo_vectors.bin")
}
