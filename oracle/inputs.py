"""Input normalisation of the reference's dataset, restated (numpy, CPU).  TEST INFRASTRUCTURE ONLY.

[ref: data/datasets/imagetext.py:131-135]
    image = image.astype('float32'); image -= image.min(); image /= image.max()
    image = torch.tensor((image - self.mean) / self.std, dtype=torch.float32)
``mean`` / ``std`` are Python floats, so every step stays float32 (numpy's weak-scalar promotion).  The lines sit inside
``ImageTextDataset.__getitem__`` behind file I/O, so they cannot be imported in isolation: parity unpinned by a
reference-generated vector; the restatement is the three lines above verbatim in operation order."""
import numpy as np


def normalize_u8(image_u8: np.ndarray, mean: float, std: float) -> np.ndarray:
    """one image, uint8 [H, W, 3] -> float32 [H, W, 3]"""
    image = image_u8.astype("float32")
    image -= image.min()
    image /= image.max()
    return ((image - mean) / std).astype(np.float32)
