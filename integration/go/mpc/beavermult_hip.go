//go:build hip

// Drop-in bodies of the local Beaver products of mpc/beavermult.go:108-147 (rows B2-B3 of SURVEY.md §8).  The originals move
// behind `//go:build !hip`; BeaverMult (scalars, :94-106) stays as it is.  NOT COMPILED in the sfgwas-hip repository.
//
// mpc-core's element types are not visible from the reference tree (un-vendored module), so the conversion between
// mpc_core.RElem and little-endian 64-bit limbs is written against the only methods the reference itself uses on them:
// Type().Modulus() / ModBitLength() and the byte (de)serialisation of the network layer (ToBytes / FromBytes,
// mpc/netconnect.go).  A maintainer with the module at hand replaces toLimbs / fromLimbs by direct field access.
package mpc

import (
	"encoding/binary"
	"math/big"

	mpc_core "github.com/hhcho/mpc-core"
	"github.com/hhcho/sfgwas/hip"
)

func limbsOf(t mpc_core.RElem) int { return int(t.ModBitLength()+63) / 64 }

func bigToLimbs(x *big.Int, limbs int, dst []uint64) {
	b := x.Bytes() // big-endian
	for i := range dst[:limbs] {
		dst[i] = 0
	}
	for i := 0; i < len(b); i++ {
		dst[i/8] |= uint64(b[len(b)-1-i]) << (8 * uint(i%8))
	}
}

func toLimbs(m mpc_core.RMat, limbs int) []uint64 {
	nr, nc := m.Dims()
	out := make([]uint64, nr*nc*limbs)
	buf := make([]byte, m.Type().NumBytes())
	for i := 0; i < nr; i++ {
		for j := 0; j < nc; j++ {
			m[i][j].ToBytes(buf) // little-endian words in mpc-core's LElem128 / LElem256
			for k := 0; k < limbs; k++ {
				out[(i*nc+j)*limbs+k] = binary.LittleEndian.Uint64(buf[8*k:])
			}
		}
	}
	return out
}

func fromLimbs(t mpc_core.RElem, flat []uint64, nr, nc, limbs int) mpc_core.RMat {
	out := mpc_core.InitRMat(t.Zero(), nr, nc)
	buf := make([]byte, t.NumBytes())
	for i := 0; i < nr; i++ {
		for j := 0; j < nc; j++ {
			for k := 0; k < limbs; k++ {
				binary.LittleEndian.PutUint64(buf[8*k:], flat[(i*nc+j)*limbs+k])
			}
			out[i][j] = t.FromBytes(buf)
		}
	}
	return out
}

func modulusLimbs(t mpc_core.RElem, limbs int) []uint64 {
	mod := make([]uint64, limbs)
	bigToLimbs(t.Modulus(), limbs, mod)
	return mod
}

// BeaverMultElemMat - mpc/beavermult.go:112-133: pid 0: am*bm; else ar*bm + br*am (+ ar*br if pid == 1), element-wise.
func (mpcObj *MPC) BeaverMultElemMat(ar, am, br, bm mpc_core.RMat) mpc_core.RMat {
	pid := mpcObj.Network.pid
	nr, nc := am.Dims()
	t := am.Type()
	limbs := limbsOf(t)
	var far, fbr []uint64
	if pid != 0 { // the dealer (pid 0) holds no shares: ar / br are unused there (:116-120)
		far, fbr = toLimbs(ar, limbs), toLimbs(br, limbs)
	}
	out := hip.Default.BeaverElem(pid, limbs, modulusLimbs(t, limbs), far, toLimbs(am, limbs), fbr, toLimbs(bm, limbs), nr*nc)
	return fromLimbs(t, out, nr, nc, limbs)
}

// BeaverMultElemVec - :108-110
func (mpcObj *MPC) BeaverMultElemVec(ar, am, br, bm mpc_core.RVec) mpc_core.RVec {
	return mpcObj.BeaverMultElemMat(mpc_core.RMat{ar}, mpc_core.RMat{am}, mpc_core.RMat{br}, mpc_core.RMat{bm})[0]
}

// BeaverMultMat - mpc/beavermult.go:135-147: pid 0: am x bm; else ar x bm + am x br (+ ar x br if pid == 1), dense products.
func (mpcObj *MPC) BeaverMultMat(ar, am, br, bm mpc_core.RMat) mpc_core.RMat {
	pid := mpcObj.Network.pid
	m, k := am.Dims()
	_, n := bm.Dims()
	t := am.Type()
	limbs := limbsOf(t)
	var far, fbr []uint64
	if pid != 0 {
		far, fbr = toLimbs(ar, limbs), toLimbs(br, limbs)
	}
	out := hip.Default.BeaverMatmul(pid, limbs, modulusLimbs(t, limbs), far, toLimbs(am, limbs), fbr, toLimbs(bm, limbs), m, k, n)
	return fromLimbs(t, out, m, n, limbs)
}
