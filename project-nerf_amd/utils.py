"""Small host utilities (reference src/utils.py)."""
import numpy as np
import torch


def compute_psnr(mse):
    """10 log10(1 / mse) for images in [0, 1] (reference src/utils.py:12-22)."""
    return 10 * np.log10(1.0 / mse)


def compute_psnr_torch(pred, target):
    return compute_psnr(torch.mean((pred - target) ** 2).item())


def render_image_safe(render_fn, model, rays_o, rays_d, near, far, n_samples, chunk, white_bkgd):
    """Halve the chunk on device OOM, down to 1024 rays (reference src/utils.py:39-76)."""
    size = int(chunk)
    while True:
        try:
            return render_fn(model=model, rays_o=rays_o, rays_d=rays_d, near=near, far=far,
                             n_samples=n_samples, chunk=size, white_bkgd=white_bkgd)
        except torch.cuda.OutOfMemoryError:
            if size <= 1024:
                raise
            torch.cuda.empty_cache()
            size = max(size // 2, 1024)
            print(f">>> device OOM, render chunk -> {size}")


def get_exp_name(cfg):
    from datetime import datetime
    return cfg.get("exp_name", datetime.now().strftime("%Y%m%d_%H%M%S"))


class TensorBoardLogger:
    """Scalar logger that degrades to a no-op without tensorboard (reference src/utils.py:86-111)."""

    def __init__(self, log_dir):
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.writer, self.enabled = SummaryWriter(log_dir), True
        except ImportError:
            self.writer, self.enabled = None, False

    def log_scalar(self, tag, value, step):
        if self.enabled:
            self.writer.add_scalar(tag, value, step)

    def log_scalars(self, main_tag, tag_scalar_dict, step):
        if self.enabled:
            self.writer.add_scalars(main_tag, tag_scalar_dict, step)

    def close(self):
        if self.enabled:
            self.writer.close()
