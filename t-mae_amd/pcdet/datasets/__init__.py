"""pcdet.datasets surface (pcdet/datasets/__init__.py:12-91): the dataset registry and build_dataloader.  The
implementation is tmae_amd.data: raw scans are read on the host, everything else runs on the device."""
from tmae_amd.data import ONCETemporalDataset, build_dataloader  # noqa: F401

__all__ = {
    'ONCETemporalDataset': ONCETemporalDataset,
}
