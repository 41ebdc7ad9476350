"""Input normalisation of the reference's dataset, restated (numpy, CPU).  TEST INFRASTRUCTURE ONLY.

[ref: data/datasets/imagetext.py:131-135]
    image = image.astype('float32'); image -= image.min(); image /= image.max()
    image = torch.tensor((image - self.mean) / self.std, dtype=torch.float32)
``mean`` / ``std`` are Python floats, so every step stays float32 (numpy's weak-scalar promotion).
Pinned: ``tests/golden/input_pipeline.npz`` holds raw uint8 pixels and the float32 batch the reference's own
``ImageTextDataset.__getitem__`` + ``collate_fn`` + trainer permute produced from them (make_golden.py
``gen_input_pipeline``); ``tests/test_oracle_golden.py::test_input_pipeline_vs_reference`` holds this function to
those vectors BIT for bit."""
import numpy as np


def normalize_u8(image_u8: np.ndarray, mean: float, std: float) -> np.ndarray:
    """one image, uint8 [H, W, 3] -> float32 [H, W, 3]"""
    image = image_u8.astype("float32")
    image -= image.min()
    image /= image.max()
    return ((image - mean) / std).astype(np.float32)
