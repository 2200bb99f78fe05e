"""tstwo_amd — MI355X (gfx950) backend for tstwo's data-parallel Circle-STARK hot path.

Host-side mirror of the reference's Backend / ColumnOps / PolyOps / FriOps / MerkleOps / QuotientOps surface
over the C ABI of libtstwo_hip.so (include/tstwo_hip.h).  The product path is GPU-only: it fails loudly when
the HIP library is missing or no GPU is present, and never touches oracle/.
"""
from . import _lib  # noqa: F401
from .backend import HipBackend, HipColumn, SecureColumnByCoords, shard_columns  # noqa: F401
from .channel import Blake2sChannel, DeviceChannel, HipGrindOps, grind  # noqa: F401
from .circle import (CanonicCoset, CircleDomain, CirclePoint, CirclePointIndex, Coset, LineDomain,  # noqa: F401
                     M31_CIRCLE_GEN, SECURE_FIELD_CIRCLE_GEN, bit_reverse_index)
from .fields import CM31, M31, P, QM31  # noqa: F401
from .fri_prover import (FriCommitPlan, FriConfig, FriLayerProof, FriProof, FriProver, LinePoly, line_interpolate,  # noqa: F401
                         compute_decommitment_positions_and_witness_evals)
from .fri_verifier import (CirclePolyDegreeBound, FriVerificationError, FriVerifier, LinePolyDegreeBound,  # noqa: F401
                           SparseEvaluation, accumulate_line, compute_decommitment_positions_and_rebuild_evals)
from .queries import Queries, get_query_positions_by_log_size  # noqa: F401
from .fri_sharded import ShardedFriLayer, fri_commit_row_sharded  # noqa: F401
from .fri import HipFriOps, decompose, fold_circle_into_line, fold_line  # noqa: F401
from .pcs import (CommitmentSchemeProof, CommitmentSchemeProver, CommitmentTreeProver, PcsConfig, PointSample,  # noqa: F401
                  TreeBuilder, column_sample_batches, compute_fri_quotients)
from .pcs_verifier import (CommitmentSchemeVerifier, VerificationError, accumulate_row_quotients,  # noqa: F401
                           fri_answers)
from .poly import (CosetSubEvaluation, HipCircleEvaluation, HipCirclePoly, LineEvaluation, SecureCirclePoly, SecureEvaluation, TwiddleTree,
                   domain_line_twiddles_from_tree, get_twiddle_dbls,  # noqa: F401
                   evaluate_polynomials, interpolate_columns, precompute_twiddles)
from .quotients import (ColumnSampleBatch, accumulate, accumulateQuotients, generate_secure_powers,  # noqa: F401
                        quotientConstants)
from .vcs import (Blake2sMerkleHasher, DeviceHashLayer, HipMerkleOps, MerkleDecommitment, MerkleProver,  # noqa: F401
                  MerkleVerifier)

__all__ = [n for n in dir() if not n.startswith("_")]
from .semantics import get_semantics, set_semantics  # noqa: F401,E402
