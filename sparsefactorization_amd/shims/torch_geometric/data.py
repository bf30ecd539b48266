"""``torch_geometric.data.DataLoader`` as the reference uses it: over a ``torch.utils.data.Dataset`` of (tensor, label)
pairs with ``batch_size, shuffle, drop_last, num_workers`` (psf_training.py:80-114). PyG 1.7's loader is a
``torch.utils.data.DataLoader`` subclass whose collate function falls through to the default one for such items."""
import torch.utils.data as _tud


class DataLoader(_tud.DataLoader):
    def __init__(self, dataset, batch_size=1, shuffle=False, follow_batch=None, exclude_keys=None, **kwargs):
        kwargs.pop("collate_fn", None)  # PyG's signature: its own collater is always used
        super().__init__(dataset, batch_size, shuffle, **kwargs)


__all__ = ["DataLoader"]
