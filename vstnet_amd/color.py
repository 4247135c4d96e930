"""Lab luminance-preserving post-process of the delldu fork (project/image_style/vstnet.py:189-220): keep the content
image's L channel, take a/b from the stylised image.  One pointwise HIP kernel (csrc/color.hip) behind
vst_lab_luminance; there is no CPU path."""
import torch

from . import _lib


def luminance_transfer(content: torch.Tensor, stylized: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """content, stylized: [B,3,H,W] fp32 on the same GPU, values in [0,1] (stylized is clamped like the fork's decoder
    output, vstnet.py:322).  Returns lab2rgb(cat(rgb2lab(content)[:, :1], rgb2lab(stylized)[:, 1:])) (color.py:94-113)."""
    if content.dim() != 4 or content.shape[1] != 3 or content.shape != stylized.shape:
        raise ValueError(f"expected two [B,3,H,W] tensors of equal shape, got {tuple(content.shape)} and {tuple(stylized.shape)}")
    if not content.is_cuda or content.device != stylized.device:
        raise RuntimeError("luminance_transfer runs on the GPU only (no CPU fallback)")
    content = content.float().contiguous()
    stylized = stylized.float().contiguous()
    if out is None:
        out = torch.empty_like(stylized)
    elif out.shape != stylized.shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != content.device:
        raise ValueError("out must be a contiguous fp32 tensor shaped like the inputs on the same device")
    B, _, H, W = content.shape
    with torch.cuda.device(content.device):
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().vst_lab_luminance(content.data_ptr(), stylized.data_ptr(), out.data_ptr(), B, H, W, stream),
                   "vst_lab_luminance")
    return out
