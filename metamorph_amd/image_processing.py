"""Image pre-processing of the SigLIP tower without the HF hub (`process_images`).

The reference takes `AutoProcessor.from_pretrained("google/siglip-so400m-patch14-384").image_processor` and overrides its
crop size with 384 x 384 (reference siglip_encoder.py:113-125); `mm_utils.process_images` then calls `.preprocess(image,
return_tensors='pt')['pixel_values']` per image (aspect ratio 'pad') or the processor on the whole list (mm_utils.py:172-188).
That checkpoint's preprocessor_config is: RGB, resize to 384 x 384 with PIL bicubic (no aspect preservation, no crop),
rescale by 1/255, normalise with mean = std = 0.5.  This class restates it (caller-side CPU work, as in the reference) so that a
tower built from a local state dict or random weights still carries an `image_processor`; tests/test_host_logic.py pins it to
transformers' own SiglipImageProcessor.
"""
from __future__ import annotations

import numpy as np
import torch


class SiglipImageProcessor:
    model_input_names = ["pixel_values"]

    def __init__(self, size: int = 384, image_mean=(0.5, 0.5, 0.5), image_std=(0.5, 0.5, 0.5), rescale_factor: float = 1.0 / 255.0):
        self.size = {"height": size, "width": size}
        self.crop_size = {"height": size, "width": size}
        self.image_mean = tuple(image_mean)
        self.image_std = tuple(image_std)
        self.rescale_factor = rescale_factor
        self.do_resize = self.do_rescale = self.do_normalize = self.do_convert_rgb = True

    def _one(self, image) -> np.ndarray:
        from PIL import Image
        if isinstance(image, torch.Tensor):
            image = image.detach().cpu().numpy()
        if isinstance(image, np.ndarray):
            if image.ndim == 3 and image.shape[0] in (1, 3) and image.shape[-1] not in (1, 3):
                image = image.transpose(1, 2, 0)                         # CHW -> HWC
            if image.dtype != np.uint8:
                raise TypeError("array images must be uint8 (HWC or CHW); pass PIL images for anything else")
            image = Image.fromarray(image.squeeze(-1) if image.shape[-1] == 1 else image)
        image = image.convert("RGB")
        image = image.resize((self.size["width"], self.size["height"]), resample=Image.BICUBIC)
        x = np.asarray(image).astype(np.float32) * np.float32(self.rescale_factor)
        x = (x - np.asarray(self.image_mean, dtype=np.float32)) / np.asarray(self.image_std, dtype=np.float32)
        return np.ascontiguousarray(x.transpose(2, 0, 1))

    def preprocess(self, images, return_tensors=None, **unused):
        if not isinstance(images, (list, tuple)):
            images = [images]
        arr = [self._one(im) for im in images]
        if return_tensors == "pt":
            return {"pixel_values": torch.from_numpy(np.stack(arr, axis=0))}
        if return_tensors in (None, "np"):
            return {"pixel_values": arr if return_tensors is None else np.stack(arr, axis=0)}
        raise ValueError(f"Unsupported tensor type: {return_tensors}")

    __call__ = preprocess
