"""zvdb_amd -- MI355X-native distance / neighbour-scan / top-k engine behind zvdb's HNSW API.

    from zvdb_amd import HNSW            # drop-in mirror of zvdb.zig's `HNSW` (src/zvdb.zig:1)
    h = HNSW(16, 200); h.insert(p); h.search(q, 10)

The compute path is libzvdb_hip.so (hand-written HIP for gfx950, C ABI in include/zvdb_hip.h).
There is no CPU fallback: importing the bindings without the built library raises."""
from ._lib import (BF_BF16, BF_F32, BF_FP4, BF_I8, ENTRY_DESCEND, FLAG_BF16_MIRROR, FLAG_BF16_ROWS, METRIC_COSINE, METRIC_DOT, METRIC_L2SQ, OPT_CERT_FORCE_REPAIR, OPT_EF_SEARCH_WIDTH,
                   OPT_CERT_MIN_LEVEL, OPT_CERTIFY, ZvdbError, set_tuning, tuned,
                   ZvdbPanic, device_count)
from .hnsw import HNSW, Node
from .index import Index, fill_synthetic_dev, merge_topk_dev, merge_topk_packed_dev, packed_result_bytes

__all__ = ["HNSW", "Node", "Index", "ZvdbError", "ZvdbPanic", "device_count", "fill_synthetic_dev",
           "merge_topk_dev", "merge_topk_packed_dev", "packed_result_bytes", "METRIC_L2SQ", "METRIC_DOT", "METRIC_COSINE", "BF_F32", "BF_BF16", "BF_I8", "BF_FP4",
           "FLAG_BF16_MIRROR", "FLAG_BF16_ROWS", "ENTRY_DESCEND", "OPT_CERTIFY", "OPT_CERT_MIN_LEVEL", "OPT_CERT_FORCE_REPAIR", "OPT_EF_SEARCH_WIDTH", "set_tuning", "tuned"]
