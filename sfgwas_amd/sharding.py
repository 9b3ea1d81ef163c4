"""SNP-block sharding arithmetic shared by bench.py and the tests.

The genotype matrix X (n_ind x m_snp) is split by blocks of 8192 SNP columns across ranks.
  Q * X    : rank r owns output block columns [blk0, blk1)            -> no collective
  Q' * X^T : rank r owns contraction block rows  [blk0, blk1)         -> reduce-scatter of the accumulators over the giant
             axis (padded to world * giants_per_rank slots), each rank aligns its giant steps, and the aligned partial
             outputs are all-reduced
"""
SLOTS = 8192
D = 91


def ceil_div(a, b):
    return (a + b - 1) // b


def snp_block_range(m_snp, rank, world, slots=SLOTS):
    """block-column range [blk0, blk1) and column range [c0, c1) of X owned by `rank`"""
    nblk = ceil_div(m_snp, slots)
    blk0, blk1 = (nblk * rank) // world, (nblk * (rank + 1)) // world
    return blk0, blk1, blk0 * slots, min(blk1 * slots, m_snp)


def giant_range(rank, world, d=D):
    return (d * rank) // world, (d * (rank + 1)) // world


def giant_slots(rank, world, d=D):
    """giant-axis sharding for the reduce-scatter: (giants per rank after padding d up to a multiple of world,
    first giant of `rank`, one past its last real giant)"""
    gpr = ceil_div(d, world)
    lo = rank * gpr
    return gpr, lo, min(lo + gpr, d)
