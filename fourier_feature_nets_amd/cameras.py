"""Camera model and camera rigs (host side).

``Resolution`` / ``CameraInfo`` keep the field names and method names the reference's scripts
use (camera_info.py:18-118).  The 4x4 pixel->world matrix is formed on the host with numpy
float32 ``inv`` exactly as the reference does (camera_info.py:66-70) so that ray directions
agree to the last ulp or two; the per-pixel work happens in kernel K1.
"""

from typing import List, NamedTuple

import numpy as np


class Resolution(NamedTuple("Resolution", [("width", int), ("height", int)])):
    """Image size in pixels."""

    def scale_to_height(self, height: int) -> "Resolution":
        return Resolution(self.width * height // self.height, height)

    def square(self) -> "Resolution":
        side = min(self.width, self.height)
        return Resolution(side, side)

    @property
    def ratio(self) -> float:
        return self.width / self.height


Ray = NamedTuple("Ray", [("origin", np.ndarray), ("direction", np.ndarray)])


class CameraInfo(NamedTuple("CameraInfo", [("name", str), ("resolution", Resolution),
                                           ("intrinsics", np.ndarray),
                                           ("extrinsics", np.ndarray)])):
    """Pinhole camera: 3x3 intrinsics and a 4x4 camera-to-world pose."""

    @staticmethod
    def create(name: str, resolution: Resolution, intrinsics: np.ndarray,
               extrinsics: np.ndarray) -> "CameraInfo":
        return CameraInfo(name, resolution, intrinsics[:3, :3], extrinsics)

    def unprojection(self) -> np.ndarray:
        """inv([[K,0],[0,1]] @ inv(E)) in float32, through the same two LAPACK inversions
        the reference performs."""
        proj = np.eye(4, dtype=np.float32)
        proj[:3, :3] = self.intrinsics
        proj = proj @ np.linalg.inv(self.extrinsics)
        return np.linalg.inv(proj)

    @property
    def position(self) -> np.ndarray:
        return self.extrinsics[:3, 3].reshape(1, 3)

    @property
    def fov_y_degrees(self) -> float:
        half = (0.5 * self.resolution.width) / self.intrinsics[1, 1]
        return 2 * np.arctan(half) * 180 / np.pi

    def project(self, positions: np.ndarray) -> np.ndarray:
        """World points -> pixel coordinates (host-side helper for callers; not on the
        rendering path)."""
        proj = np.eye(4, dtype=np.float32)
        proj[:3, :3] = self.intrinsics
        proj = proj @ np.linalg.inv(self.extrinsics)
        homog = np.concatenate([positions, np.ones((positions.shape[0], 1), np.float32)], -1)
        pts = (proj @ homog.T).T
        return pts[:, :2] / pts[:, 2:3]

    def raycast(self, points: np.ndarray, device="cuda") -> Ray:
        """Origins and unit directions of the rays through 2-D pixel positions (kernel K1
        with an explicit point list).  Returns float32 numpy arrays like the reference."""
        import torch
        from . import ops
        pts = torch.as_tensor(np.ascontiguousarray(points, dtype=np.float32)).reshape(-1, 2)
        dev = torch.device(device)
        unproj = torch.from_numpy(self.unprojection().astype(np.float32)).reshape(1, 4, 4).to(dev)
        cam = torch.from_numpy(self.position.astype(np.float32)).to(dev)
        big = 1e30
        starts, dirs, _, _ = ops.raygen_nearfar(unproj.contiguous(), cam.contiguous(),
                                                pts.shape[0], 1, [-big] * 3, [big] * 3,
                                                pts.to(dev).contiguous())
        return Ray(starts.cpu().numpy(), dirs.cpu().numpy())


def _axis_angle(axis: np.ndarray, angle: float) -> np.ndarray:
    """4x4 rotation about a unit axis (Rodrigues)."""
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    x, y, z = axis
    c, s = np.cos(angle), np.sin(angle)
    k = 1 - c
    rot = np.array([[c + x * x * k, x * y * k - z * s, x * z * k + y * s, 0],
                    [y * x * k + z * s, c + y * y * k, y * z * k - x * s, 0],
                    [z * x * k - y * s, z * y * k + x * s, c + z * z * k, 0],
                    [0, 0, 0, 1]])
    return rot


def _look_at_camera_to_world(eye: np.ndarray, target: np.ndarray, up: np.ndarray) -> np.ndarray:
    """Camera-to-world of a computer-vision camera (+z forward, +y down) at ``eye``."""
    forward = np.asarray(target, np.float64) - np.asarray(eye, np.float64)
    forward = forward / np.linalg.norm(forward)
    right = np.cross(forward, np.asarray(up, np.float64))
    right = right / np.linalg.norm(right)
    down = np.cross(forward, right)
    pose = np.eye(4)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, down, forward, eye
    return pose


def orbit(up_dir: np.ndarray, forward_dir: np.ndarray, num_frames: int, fov_y_degrees: float,
          resolution: Resolution, distance: float, min_altitude=np.pi / 12,
          max_altitude=np.pi / 4) -> List[CameraInfo]:
    """Two revolutions around the origin with the altitude ramping up and back down.

    Same call signature and intent as the reference's ``utils.orbit`` (utils.py:244-303),
    written without scenepic: base pose = a look-at camera at ``-forward_dir*distance``;
    frame pose = R(up, azimuth) @ R(right, altitude) @ base with right = up x forward.
    (The reference builds the base pose through scenepic, which is not available; poses are
    therefore "parity unpinned" -- see DESIGN.md.)
    """
    up_dir = np.asarray(up_dir, np.float64)
    forward_dir = np.asarray(forward_dir, np.float64)
    right_dir = np.cross(up_dir, forward_dir)
    azimuth = np.linspace(0, 4 * np.pi, num_frames, endpoint=False)
    altitude = np.zeros_like(azimuth)
    half = num_frames // 2
    altitude[:half] = np.linspace(min_altitude, max_altitude, half, endpoint=False)
    altitude[half:] = np.linspace(max_altitude, min_altitude, num_frames - half, endpoint=False)
    focal = .5 * resolution.width / np.tan(.5 * fov_y_degrees * np.pi / 180)
    intrinsics = np.array([focal, 0, resolution.width / 2,
                           0, focal, resolution.height / 2,
                           0, 0, 1], np.float32).reshape(3, 3)
    base = _look_at_camera_to_world(-forward_dir * distance, np.zeros(3), up_dir)
    cameras = []
    for azi, alt in zip(azimuth, altitude):
        pose = _axis_angle(up_dir, azi) @ _axis_angle(right_dir, alt) @ base
        cameras.append(CameraInfo.create("cam%d" % len(cameras), resolution, intrinsics,
                                         pose.astype(np.float32)))
    return cameras
