"""Blender-format dataset (reference src/dataset.py:9-171) plus a synthetic scene writer: no
dataset is available offline, so measurements use an analytic scene in the same file format."""
import json
import os

import numpy as np
import torch


class BlenderDataset:
    """transforms_{split}.json + RGBA frames -> rays.  Host-side I/O; ray maths follows the
    reference (no +0.5 pixel centre, -y, -z, normalised directions, scene_scale on origins only)."""

    def __init__(self, root_dir, split="train", downscale=1, white_bkgd=True, scene_scale=1.0):
        from PIL import Image
        self.root_dir, self.split = root_dir, split
        self.downscale = max(int(downscale), 1)
        self.white_bkgd, self.scene_scale = white_bkgd, float(scene_scale)
        with open(os.path.join(root_dir, f"transforms_{split}.json"), "r", encoding="utf-8") as f:
            meta = json.load(f)
        self.camera_angle_x = float(meta["camera_angle_x"])
        self.frames = meta["frames"]
        images, poses = [], []
        for frame in self.frames:
            path = frame["file_path"]
            path = path[2:] if path.startswith("./") else path
            path = os.path.join(root_dir, path)
            if not os.path.splitext(path)[1]:
                path += ".png" if os.path.exists(path + ".png") else ".jpg"
            img = Image.open(path).convert("RGBA")
            if self.downscale > 1:
                img = img.resize((img.width // self.downscale, img.height // self.downscale), Image.LANCZOS)
            images.append(torch.from_numpy(np.array(img).astype(np.float32) / 255.0))
            poses.append(torch.tensor(frame["transform_matrix"], dtype=torch.float32))
        self.images, self.poses = torch.stack(images, 0), torch.stack(poses, 0)
        self.H, self.W = self.images.shape[1:3]
        self.focal = 0.5 * self.W / np.tan(0.5 * self.camera_angle_x)
        self._directions = self._build_directions()

    def _build_directions(self):
        j, i = torch.meshgrid(torch.arange(self.H), torch.arange(self.W), indexing="ij")
        return torch.stack([(i - self.W * 0.5) / self.focal, -(j - self.H * 0.5) / self.focal,
                            -torch.ones_like(i)], dim=-1)

    def __len__(self):
        return self.images.shape[0]

    def get_rays(self, c2w):
        d = torch.matmul(self._directions.to(c2w.device).reshape(-1, 3), c2w[:3, :3].T).reshape(self.H, self.W, 3)
        d = d / torch.norm(d, dim=-1, keepdim=True)
        o = c2w[:3, 3].expand_as(d)
        if self.scene_scale != 1.0:
            o = o * self.scene_scale
        return o, d

    def get_image_rays(self, index, device):
        o, d = self.get_rays(self.poses[index])
        rgba = self.images[index]
        rgb, a = rgba[..., :3], rgba[..., 3:4]
        target = rgb * a + (1.0 - a) if self.white_bkgd else rgb * a
        return o.to(device), d.to(device), target.to(device)

    def to(self, device):
        """Keep frames and poses resident on ``device`` so that batch sampling needs no host work
        (the reference samples on the CPU and copies three tensors per step)."""
        self.images, self.poses = self.images.to(device), self.poses.to(device)
        self._directions = self._directions.to(device)
        return self

    def sample_batch(self, batch_size, bg, shard=None):
        """Training batch from GPU-resident frames: one uniform draw over all pixels of all frames (the same
        distribution as the reference's three draws, dataset.py:147-150) and one kernel that forms the rays
        AND the composited target rgb * a + bg * (1 - a) (run.py:317-322).  ``shard`` = (lo, hi): every data-parallel
        rank draws the SAME ``batch_size`` pixels (same torch seed on every rank) and forms rays [lo, hi) of them --
        the ranks' batches are the shards of one global batch."""
        from . import ops
        idx = torch.randint(0, len(self) * self.H * self.W, (batch_size,), device=self.images.device)
        if shard is not None:
            idx = idx[shard[0]:shard[1]].contiguous()
        o, d, target, _ = ops.gather_batch(self.images, self.poses, idx, self.focal, self.scene_scale, bg=bg)
        return o, d, target

    def train_batch(self, batch_size, n_samples, near, far, bg, seed, counter, perturb=True, first_ray=0):
        """(rays_o, rays_d, target, z) of one training step from one kernel (ops.train_batch): pixel draws,
        rays, composited targets and jittered stratified depths.  ``first_ray`` = rank * batch_size with the same seed and
        counter on every rank: the ranks' batches are then the shards of ONE global batch of world * batch_size rays."""
        from . import ops
        return ops.train_batch(self.images, self.poses, self.focal, batch_size, n_samples, near, far, seed, counter, bg=bg,
                               scene_scale=self.scene_scale, perturb=perturb, first_ray=first_ray)

    @classmethod
    def from_tensors(cls, images, poses, camera_angle_x, white_bkgd=True, scene_scale=1.0):
        """Dataset over frames that are already tensors ([n,H,W,4] RGBA in [0,1], [n,4,4] camera-to-world)."""
        self = cls.__new__(cls)
        self.root_dir, self.split, self.downscale = None, "memory", 1
        self.white_bkgd, self.scene_scale = white_bkgd, float(scene_scale)
        self.camera_angle_x = float(camera_angle_x)
        self.frames = [None] * images.shape[0]
        self.images, self.poses = images, poses
        self.H, self.W = images.shape[1:3]
        self.focal = 0.5 * self.W / np.tan(0.5 * self.camera_angle_x)
        self._directions = self._build_directions().to(images.device)
        return self

    def sample_random_rays(self, batch_size, device):
        dev = self.images.device
        img = torch.randint(0, len(self), (batch_size,), device=dev)
        py = torch.randint(0, self.H, (batch_size,), device=dev)
        px = torch.randint(0, self.W, (batch_size,), device=dev)
        if dev.type == "cuda" and dev == torch.device(device):
            # frames resident on the GPU (``.to(device)``): one kernel instead of a batched 3x3 GEMM
            # and a dozen elementwise launches (0.22 ms of a 1.5 ms Instant-NGP step)
            from . import ops
            return ops.gather_rays(self.images, self.poses, img, py, px, self.focal, self.scene_scale)
        c2w = self.poses[img]
        dirs = torch.stack([(px - self.W * 0.5) / self.focal, -(py - self.H * 0.5) / self.focal,
                            -torch.ones_like(px)], dim=-1)
        d = torch.bmm(c2w[:, :3, :3], dirs.unsqueeze(-1)).squeeze(-1)
        o = c2w[:, :3, 3]
        if self.scene_scale != 1.0:
            o = o * self.scene_scale
        rgba = self.images[img, py, px]
        d = d / torch.norm(d, dim=-1, keepdim=True)
        return o.to(device), d.to(device), rgba.to(device)


class DynamicDataset(BlenderDataset):
    """Dynamic scene (reference src/dataset.py:174-294): every frame carries a time stamp in [0, 1] (``time`` in
    the json, else its position in the sequence); rgb and alpha are kept apart, ``images`` is the composited
    target.  get_image_rays -> (o, d, target, time [1,1]); sample_random_rays -> (o, d, rgba [B,4], times [B,1])."""

    def __init__(self, root_dir, split="train", downscale=1, white_bkgd=True, scene_scale=1.0):
        super().__init__(root_dir, split, downscale, white_bkgd, scene_scale)
        n = len(self.frames)
        self.times = torch.tensor([float(f["time"]) if "time" in f else (i / (n - 1) if n > 1 else 0.0)
                                   for i, f in enumerate(self.frames)], dtype=torch.float32)
        self.images_rgb, self.images_alpha = self.images[..., :3].contiguous(), self.images[..., 3:4].contiguous()
        self.rgba = self.images
        self.images = self.images_rgb * self.images_alpha + ((1.0 - self.images_alpha) if white_bkgd else 0.0)

    def to(self, device):
        self.rgba, self.poses, self.times = self.rgba.to(device), self.poses.to(device), self.times.to(device)
        self.images, self.images_rgb, self.images_alpha = self.images.to(device), self.images_rgb.to(device), self.images_alpha.to(device)
        self._directions = self._directions.to(device)
        return self

    def get_image_rays(self, index, device):
        o, d = self.get_rays(self.poses[index])
        return o.to(device), d.to(device), self.images[index].to(device), self.times[index].view(1, 1).to(device)

    def sample_random_rays(self, batch_size, device):
        dev = self.rgba.device
        img = torch.randint(0, len(self), (batch_size,), device=dev)
        py = torch.randint(0, self.H, (batch_size,), device=dev)
        px = torch.randint(0, self.W, (batch_size,), device=dev)
        times = self.times[img].unsqueeze(-1)
        if dev.type == "cuda" and dev == torch.device(device):
            from . import ops
            o, d, rgba = ops.gather_rays(self.rgba, self.poses, img, py, px, self.focal, self.scene_scale)
            return o, d, rgba, times
        c2w = self.poses[img]
        dirs = torch.stack([(px - self.W * 0.5) / self.focal, -(py - self.H * 0.5) / self.focal, -torch.ones_like(px)], dim=-1)
        d = torch.bmm(c2w[:, :3, :3], dirs.unsqueeze(-1)).squeeze(-1)
        o = c2w[:, :3, 3]
        if self.scene_scale != 1.0:
            o = o * self.scene_scale
        d = d / torch.norm(d, dim=-1, keepdim=True)
        return o.to(device), d.to(device), self.rgba[img, py, px].to(device), times.to(device)


# ---------------------------------------------------------------------------------------------
def look_at_pose(eye):
    """Camera-to-world with -z looking at the origin, z-up world (Blender convention)."""
    eye = np.asarray(eye, np.float64)
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    right /= np.linalg.norm(right) + 1e-12
    up = np.cross(right, fwd)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up, -fwd, eye
    return m


def analytic_scene(pts):
    """Union of a box and two spheres inside [-1,1]^3: returns (rgb [N,3], sigma [N])."""
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    box = ((x.abs() < 0.9) & (y.abs() < 0.6) & ((z + 0.35).abs() < 0.4)).float()
    s1 = (((x - 0.35) ** 2 + (y + 0.1) ** 2 + (z - 0.45) ** 2) < 0.5 ** 2).float()
    s2 = (((x + 0.5) ** 2 + (y - 0.2) ** 2 + (z - 0.35) ** 2) < 0.38 ** 2).float()
    sigma = 40.0 * torch.clamp(box + s1 + s2, max=1.0)
    rgb = torch.stack([0.5 + 0.5 * torch.sin(6 * x + 1.0), 0.5 + 0.5 * torch.sin(5 * y + 2.0),
                       0.5 + 0.5 * torch.sin(7 * z)], dim=-1)
    rgb = torch.where(s1[:, None] > 0, torch.tensor([0.9, 0.2, 0.15]).to(pts) * torch.ones_like(rgb), rgb)
    return rgb, sigma


def render_analytic_frame(c2w, size, focal, n_samples=256, device="cpu", chunk=32768):
    """RGBA ground truth [size,size,4] of ``analytic_scene`` from camera ``c2w`` by plain quadrature
    (ground truth only; not a product path).  ``device`` may be a GPU: the bench's 800 x 800 frames."""
    c2w = c2w.to(device)
    j, i = torch.meshgrid(torch.arange(size, device=device), torch.arange(size, device=device), indexing="ij")
    d = torch.stack([(i - size * 0.5) / focal, -(j - size * 0.5) / focal, -torch.ones_like(i)], -1).reshape(-1, 3).float()
    d = d @ c2w[:3, :3].T
    d = d / d.norm(dim=-1, keepdim=True)
    t = torch.linspace(2.0, 6.0, n_samples, device=device)
    out = []
    for k in range(0, d.shape[0], chunk):
        dd = d[k:k + chunk]
        pts = c2w[:3, 3][None, None] + dd[:, None] * t[None, :, None]
        rgb, sigma = analytic_scene(pts.reshape(-1, 3))
        rgb, sigma = rgb.view(-1, n_samples, 3), sigma.view(-1, n_samples)
        alpha = 1 - torch.exp(-sigma * (4.0 / (n_samples - 1)))
        T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], -1), -1)[:, :-1]
        w = alpha * T
        acc = w.sum(-1, keepdim=True)
        col = (w[..., None] * rgb).sum(1) / acc.clamp_min(1e-6)
        out.append(torch.cat([col.clamp(0, 1), acc.clamp(0, 1)], -1))
    return torch.cat(out, 0).view(size, size, 4)


def synthetic_poses(count, rng):
    """Cameras on the upper hemisphere at the NeRF-Synthetic radius, looking at the origin."""
    poses = []
    for _ in range(count):
        th, ph = rng.uniform(0, 2 * np.pi), rng.uniform(0.15, 1.2)
        eye = 4.0311 * np.array([np.cos(th) * np.cos(ph), np.sin(th) * np.cos(ph), np.sin(ph)])
        poses.append(torch.tensor(look_at_pose(eye), dtype=torch.float32))
    return poses


SYNTHETIC_CAMERA_ANGLE = 0.6911112070083618


def synthetic_frames(count, size, device, seed=2025, n_samples=192):
    """``count`` RGBA frames of the analytic scene rendered ON ``device`` (no PNG round trip): images
    [count,size,size,4] in [0,1] (8-bit quantised like the files would be), poses [count,4,4]."""
    rng = np.random.default_rng(seed)
    focal = 0.5 * size / np.tan(0.5 * SYNTHETIC_CAMERA_ANGLE)
    poses = synthetic_poses(count, rng)
    frames = [torch.round(render_analytic_frame(p, size, focal, n_samples, device) * 255.0) / 255.0 for p in poses]
    return torch.stack(frames, 0), torch.stack(poses, 0).to(device)


def write_synthetic_scene(root, n_train=20, n_test=4, size=100, seed=2025, n_samples=256):
    """Writes transforms_{train,test}.json + RGBA PNGs rendered from ``analytic_scene`` by plain
    quadrature on the host (ground truth only; not a product path)."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    angle = SYNTHETIC_CAMERA_ANGLE
    focal = 0.5 * size / np.tan(0.5 * angle)
    os.makedirs(root, exist_ok=True)
    for split, count in (("train", n_train), ("test", n_test)):
        frames = []
        os.makedirs(os.path.join(root, split), exist_ok=True)
        for k, c2w in enumerate(synthetic_poses(count, rng)):
            rgba = render_analytic_frame(c2w, size, focal, n_samples)
            Image.fromarray((rgba.numpy() * 255 + 0.5).astype(np.uint8), "RGBA").save(os.path.join(root, split, f"r_{k}.png"))
            frames.append({"file_path": f"./{split}/r_{k}", "transform_matrix": c2w.tolist()})
        with open(os.path.join(root, f"transforms_{split}.json"), "w") as f:
            json.dump({"camera_angle_x": angle, "frames": frames}, f)
    return root
