"""Environment-map Blinn-Phong shading for the FIT_INVERSE task -- host mirror of the reference's
``src/utils/pytorch3d_envmap_shader.py`` over ``reni_envmap_shade`` / ``reni_envmap_shade_backward``
(include/reni_hip.h).

Same names and argument meaning as the reference: ``EnvironmentMap`` (:33-44),
``blinn_phong_shading_env_map`` (:46-116), ``BlinnPhongShaderEnvMap`` (:119-174), ``build_renderer`` (:177-217).
The shading arithmetic runs in the HIP library (there is no torch fallback: a CPU tensor raises); the only
tensor that carries a gradient is ``EnvironmentMap.environment_map`` -- the mesh, the camera and the texel grid
are constants in the reference's use (RENI_module.py:386-396).

pytorch3d (rasteriser, ``Meshes``, cameras) is not part of this build.  The shading function is duck-typed over
what it touches -- ``meshes.verts_packed() / faces_packed() / verts_normals_packed()``,
``fragments.pix_to_face / bary_coords``, ``cameras.get_camera_center()``, ``materials.shininess`` -- so real
pytorch3d objects work when the package is present, and a fixed G-buffer works without it
(``GBuffer`` / ``blinn_phong_shading_gbuffer``).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops


class EnvironmentMap:
    """Lighting colours and directions (reference :33-44): the map is pre-multiplied by the sine weight."""

    def __init__(self, environment_map: torch.Tensor = None, directions: torch.Tensor = None,
                 sineweight: torch.Tensor = None) -> None:
        self.directions = directions
        self.environment_map = environment_map * sineweight

    def to(self, device):
        self.directions = self.directions.to(device)
        self.environment_map = self.environment_map.to(device)
        return self


def interpolate_face_attributes(pix_to_face: torch.Tensor, bary_coords: torch.Tensor, face_attrs: torch.Tensor):
    """Barycentric interpolation of per-face-vertex attributes (pytorch3d.ops.interpolate_face_attributes):
    pix_to_face [N,H,W,K] (-1 = background), bary_coords [N,H,W,K,3], face_attrs [F,3,D] -> [N,H,W,K,D]."""
    mask = pix_to_face < 0
    attrs = face_attrs[pix_to_face.clamp(min=0)]
    out = (bary_coords[..., None] * attrs).sum(dim=-2)
    return out.masked_fill(mask[..., None], 0.0)


class _ShadeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, light_colors, normals, positions, camera_center, light_dirs, shininess, kd, ks):
        ctx.save_for_backward(normals, positions, camera_center, light_dirs)
        ctx.consts = (float(shininess), float(kd), float(ks))
        return ops.envmap_shade(normals, positions, camera_center, light_dirs, light_colors, *ctx.consts)

    @staticmethod
    def backward(ctx, dcolors):
        normals, positions, camera_center, light_dirs = ctx.saved_tensors
        g = ops.envmap_shade_backward(normals, positions, camera_center, light_dirs, dcolors.contiguous(), *ctx.consts)
        return g, None, None, None, None, None, None, None


def _shared_grid(directions: torch.Tensor) -> torch.Tensor:
    """[B,J,3] directions -> [J,3] when every image uses the same grid (the reference's directions.repeat(B,1,1),
    RENI_module.py:376): the shading coefficients are then evaluated once for the whole batch."""
    if directions.dim() == 2:
        return directions
    if directions.shape[0] == 1 or bool((directions == directions[:1]).all()):
        return directions[0]
    return directions


def blinn_phong_shading_gbuffer(pixel_normals, pixel_positions, camera_center, envmap: EnvironmentMap,
                                shininess, kd, ks) -> torch.Tensor:
    """colors [B, NP, 3] from interpolated (not normalised) normals / positions [NP, 3] (reference :75-115)."""
    s = float(torch.as_tensor(shininess).reshape(-1)[0])
    cam = torch.as_tensor(camera_center, dtype=torch.float32).reshape(-1)[:3].cpu()
    return _ShadeFn.apply(envmap.environment_map, pixel_normals, pixel_positions, cam, _shared_grid(envmap.directions),
                          s, kd, ks)


def blinn_phong_shading_env_map(device, meshes, fragments, envmap, cameras, materials, kd, ks):
    """Reference signature (:46-48).  Returns (colors [B,H,W,3], pixel_normals [B,H,W,3])."""
    verts = meshes.verts_packed()
    faces = meshes.faces_packed()
    vertex_normals = meshes.verts_normals_packed()
    pixel_positions = interpolate_face_attributes(fragments.pix_to_face, fragments.bary_coords, verts[faces])
    pixel_normals = interpolate_face_attributes(fragments.pix_to_face, fragments.bary_coords, vertex_normals[faces])
    if pixel_normals.shape[0] != 1 or pixel_normals.shape[3] != 1:
        raise ValueError("one mesh and one face per pixel, as in the reference (:80-81 'assume K = 1')")
    _, Hr, Wr, _, _ = pixel_normals.shape
    B = envmap.directions.shape[0]
    n_flat = pixel_normals[0, :, :, 0, :].reshape(Hr * Wr, 3).to(device)
    p_flat = pixel_positions[0, :, :, 0, :].reshape(Hr * Wr, 3).to(device)
    cam = cameras.get_camera_center().reshape(-1)[:3]
    envmap_dev = envmap
    if envmap.environment_map.device != torch.device(device) and str(envmap.environment_map.device) != str(device):
        envmap_dev = EnvironmentMap.__new__(EnvironmentMap)
        envmap_dev.directions = envmap.directions.to(device)
        envmap_dev.environment_map = envmap.environment_map.to(device)
    colors = blinn_phong_shading_gbuffer(n_flat, p_flat, cam, envmap_dev, materials.shininess, kd, ks)
    normals_out = torch.nn.functional.normalize(n_flat, p=2, dim=-1, eps=1e-6).reshape(1, Hr, Wr, 3).repeat(B, 1, 1, 1)
    return colors.reshape(B, Hr, Wr, 3), normals_out


class GBuffer:
    """A rasterised view held as data: what the reference's rasteriser + interpolation produce for its fixed mesh and
    camera (RENI_module.py:66-72).  Stands in for (meshes, fragments, cameras) where pytorch3d is absent."""

    def __init__(self, pixel_normals: torch.Tensor, pixel_positions: torch.Tensor, camera_center, image_size):
        Hr, Wr = (image_size, image_size) if isinstance(image_size, int) else image_size
        self.pixel_normals = pixel_normals.reshape(Hr * Wr, 3)
        self.pixel_positions = pixel_positions.reshape(Hr * Wr, 3)
        self.camera_center = torch.as_tensor(camera_center, dtype=torch.float32).reshape(-1)[:3]
        self.image_size = (Hr, Wr)

    def to(self, device):
        self.pixel_normals = self.pixel_normals.to(device)
        self.pixel_positions = self.pixel_positions.to(device)
        return self


class _Materials:
    def __init__(self, shininess=64.0):
        self.shininess = torch.as_tensor([float(shininess)])

    def to(self, device):
        return self


class BlinnPhongShaderEnvMap(nn.Module):
    """Per-pixel lighting by an environment map (reference :119-174)."""

    def __init__(self, device="cpu", cameras=None, envmap: EnvironmentMap = None, materials=None, kd=None, ks=None):
        super().__init__()
        self.envmap = envmap
        self.materials = materials if materials is not None else _Materials()
        self.cameras = cameras
        self.device = device
        self.kd = kd
        self.ks = ks

    def to(self, device):
        cameras = self.cameras
        if cameras is not None and hasattr(cameras, "to"):
            self.cameras = cameras.to(device)
        if hasattr(self.materials, "to"):
            self.materials = self.materials.to(device)
        if self.envmap is not None:
            self.envmap = self.envmap.to(device)
        self.device = device
        return self

    def forward(self, fragments, meshes, envmap: EnvironmentMap, **kwargs):
        cameras = kwargs.get("cameras", self.cameras)
        if cameras is None:
            raise ValueError("Cameras must be specified either at initialization or in the forward pass of BlinnPhongShader")
        materials = kwargs.get("materials", self.materials)
        return blinn_phong_shading_env_map(device=self.device, meshes=meshes, fragments=fragments, envmap=envmap,
                                           cameras=cameras, materials=materials, kd=self.kd, ks=self.ks)


class GBufferRenderer(nn.Module):
    """``renderer(envmap=...) -> (render [B,Hr,Wr,3], pixel_normals)`` over a fixed G-buffer: the call shape of the
    reference's MeshRenderer in ``RENI.get_render`` (RENI_module.py:393-396) without the rasteriser."""

    def __init__(self, gbuffer: GBuffer, kd: float, shininess: float = 500.0):
        super().__init__()
        self.gbuffer = gbuffer
        self.kd = kd
        self.ks = 1.0 - kd               # :199
        self.shininess = shininess       # Materials(shininess=500), :186

    def forward(self, envmap: EnvironmentMap = None, **kwargs):
        g = self.gbuffer
        dev = envmap.environment_map.device
        if g.pixel_normals.device != dev:
            g.to(dev)
        B = envmap.environment_map.shape[0]
        Hr, Wr = g.image_size
        colors = blinn_phong_shading_gbuffer(g.pixel_normals, g.pixel_positions, g.camera_center, envmap, self.shininess,
                                             self.kd, self.ks)
        normals = torch.nn.functional.normalize(g.pixel_normals, p=2, dim=-1, eps=1e-6).reshape(1, Hr, Wr, 3).repeat(B, 1, 1, 1)
        return colors.reshape(B, Hr, Wr, 3), normals


def build_renderer(obj_path, obj_rotation, img_size, kd, device):
    """Same signature and return value as the reference's ``build_renderer`` (:177-217): ``(renderer, R, T, mesh)`` with
    ``renderer(meshes_world=mesh, R=R, T=T, envmap=...)`` -> ``(render, pixel_normals)``.  pytorch3d does the mesh
    loading and the one rasterisation; the shading is this module's HIP path.  Without pytorch3d use
    ``GBufferRenderer`` with a stored G-buffer."""
    try:
        import pytorch3d.renderer as p3r
        from pytorch3d.io import load_obj
        from pytorch3d.structures import Meshes
        from pytorch3d.transforms import RotateAxisAngle
    except ImportError as e:  # pytorch3d is absent from this image
        raise ImportError("build_renderer needs pytorch3d (mesh loading and rasterisation); with a stored G-buffer use "
                          "reni_amd.envmap_shader.GBufferRenderer instead") from e
    # pragma: no cover -- everything below needs pytorch3d
    verts, faces_idx, _ = load_obj(obj_path, load_textures=False, device=device)
    verts = RotateAxisAngle(obj_rotation, "Y", device=device).transform_points(verts).to(device)
    white = p3r.TexturesVertex(verts_features=torch.ones(1, verts.shape[0], 3, device=device))
    mesh = Meshes(verts=[verts], faces=[faces_idx.verts_idx.to(device)], textures=white)
    cameras = p3r.FoVPerspectiveCameras(device=device)
    shader = BlinnPhongShaderEnvMap(device=device, cameras=cameras, materials=p3r.Materials(shininess=500), kd=kd, ks=1.0 - kd)
    raster = p3r.MeshRasterizer(cameras=cameras, raster_settings=p3r.RasterizationSettings(
        image_size=img_size, blur_radius=0.0, faces_per_pixel=1, perspective_correct=False))
    R, T = p3r.look_at_view_transform(2.0, 0.0, 0.0, degrees=True, device=device)
    return p3r.MeshRenderer(rasterizer=raster, shader=shader), R, T, mesh
