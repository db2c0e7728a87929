"""Host-side partitioning for the two multi-GPU modes; used by bench.py and the gloo tests.

sharded: one source cloud cut into `world` contiguous, near-equal ranges; every rank holds a map replica and the
         29 H/b/chi2/n terms are all-reduced on each Gauss-Newton iteration (RCCL inside the library).
replay : independent (scan, guess) pairs dealt round-robin; no collective on the data path.
"""

N_REDUCED = 32   # doubles per all-reduce row: 21 H + 6 b + chi2 + n_eff + hits + 2 spare


def shard_range(n, rank, world):
    """[lo, hi) of rank's contiguous share of n source points; shares differ by at most one point."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return n * rank // world, n * (rank + 1) // world


def replay_assignment(n_pairs, rank, world):
    """Indices of the scan pairs rank aligns in offline replay (round-robin)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_pairs, world))
