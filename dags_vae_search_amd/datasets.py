"""In-memory datasets of the reference's train / test drivers, igraph- and dask-free.

``LabeledDagDatasetInMemory`` (experiments/03_synthetic_12/main.py:34-52; 01_bn_asia/main.py): every parquet row
(``l{v}`` uint16 labels, ``e{v}`` '0/1' strings, src/toolkit/labeled.py:117-130) becomes one graph whose PACE features are
computed ONCE at load (``model.prepare_features([graph])``) and collated per batch by ``pace_collate_fn``.
``LabeledDagDatasetInMemoryTest`` (main.py:54-72) keeps the graphs themselves (for ``batch_test``).  The reference reads
the parquet directory with dask; here pyarrow reads it directly (``rows=`` accepts already-decoded row dicts).
For large data sets prefer ``CompactDagDataset`` (records.py): the whole set stays on the GPU as the 3-byte row codec.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

from .features import LabeledDag


def read_parquet_rows(dataset_dir: str) -> List[Dict]:
    import pyarrow.parquet as pq
    return pq.read_table(dataset_dir).to_pylist()


def write_parquet_rows(path: str, toolkit: LabeledDag, graphs: Sequence) -> None:
    """Row codec -> one parquet file with the reference's schema (labeled.py:117-130: uint16 labels, string edges)."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    rows = [toolkit.from_graph_to_dict_writable(g) for g in graphs]
    n = toolkit.num_vertices
    schema = pa.schema([pa.field(f"{toolkit.dict_label_prefix}{v}", pa.uint16()) for v in range(n)] +
                       [pa.field(f"{toolkit.dict_edges_prefix}{v}", pa.string()) for v in range(n)])
    pq.write_table(pa.Table.from_pylist(rows, schema=schema), path)


class LabeledDagDatasetInMemory:
    def __init__(self, dataset_dir: Optional[str], toolkit: LabeledDag, model, rows: Optional[Iterable[Dict]] = None):
        self.toolkit = toolkit
        self.model = model
        rows = read_parquet_rows(dataset_dir) if rows is None else list(rows)
        self.graphs = [self.model.prepare_features([self.toolkit.from_dict_to_graph(r)]) for r in rows]

    def __len__(self) -> int:
        return len(self.graphs)

    def __getitem__(self, idx: int):
        return self.graphs[idx]


class LabeledDagDatasetInMemoryTest:
    def __init__(self, dataset_dir: Optional[str], toolkit: LabeledDag, model=None, rows: Optional[Iterable[Dict]] = None):
        self.toolkit = toolkit
        self.model = model
        rows = read_parquet_rows(dataset_dir) if rows is None else list(rows)
        self.graphs = [self.toolkit.from_dict_to_graph(r) for r in rows]

    def __len__(self) -> int:
        return len(self.graphs)

    def __getitem__(self, idx: int):
        return self.graphs[idx]
