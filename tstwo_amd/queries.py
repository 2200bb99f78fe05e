"""Queries (reference src/queries.ts:20-190): sorted, de-duplicated query positions drawn from the channel."""
from __future__ import annotations

UPPER_BOUND_QUERY_BYTES = 4


class Queries:
    def __init__(self, positions, log_domain_size: int):
        if log_domain_size < 0:
            raise TypeError("logDomainSize must be a non-negative integer")
        positions = list(positions)
        for i, p in enumerate(positions):
            if p < 0 or p >= (1 << log_domain_size):
                raise TypeError(f"Invalid position at index {i}: {p}")
            if i and positions[i - 1] > p:
                raise TypeError("Positions must be sorted in ascending order")
        self.positions, self.log_domain_size = positions, log_domain_size

    @staticmethod
    def generate(channel, log_domain_size: int, n_queries: int) -> "Queries":
        """queries.ts:70-103: 4-byte little-endian words of draw_random_bytes(), masked to the domain size."""
        if log_domain_size > 31:
            raise TypeError("logDomainSize must be at most 31 for JavaScript safety")
        mask = (1 << log_domain_size) - 1
        seen = set()
        while len(seen) < n_queries:
            b = channel.draw_random_bytes()
            for i in range(0, len(b) - UPPER_BOUND_QUERY_BYTES + 1, UPPER_BOUND_QUERY_BYTES):
                seen.add(int.from_bytes(b[i:i + 4], "little") & mask)
                if len(seen) == n_queries:
                    break
        return Queries(sorted(seen), log_domain_size)

    from_positions = fromPositions = staticmethod(lambda positions, log_domain_size: Queries(positions, log_domain_size))

    def fold(self, n_folds: int) -> "Queries":
        """queries.ts:140-158."""
        if n_folds < 0:
            raise TypeError("nFolds must be a non-negative integer")
        if n_folds > self.log_domain_size:
            raise ValueError("nFolds too large")
        return Queries(sorted({q >> n_folds for q in self.positions}), self.log_domain_size - n_folds)

    def __len__(self): return len(self.positions)
    def __iter__(self): return iter(self.positions)
    def __eq__(self, o): return isinstance(o, Queries) and (self.positions, self.log_domain_size) == (o.positions, o.log_domain_size)
    length = property(lambda self: len(self.positions))


def get_query_positions_by_log_size(queries: Queries, column_log_sizes) -> dict:
    """fri.ts:470-480."""
    return {lg: list(queries.fold(queries.log_domain_size - lg).positions) for lg in sorted(set(column_log_sizes), reverse=True)}
