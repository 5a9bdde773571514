"""Materialised invariant encodings with the reference's names (src/models/RENI.py:23-60).

NOT on the product path: ``model(Z, directions)`` never builds these tensors (the fused kernels
fold them into a per-image affine map).  They exist because the reference
exposes them as ``model.InvariantRepresentation`` and they are handy for inspecting the encoding.
"""
import torch


def SO3InvariantRepresentation(Z, D):
    """[D Z^T | vec(Z Z^T)] -> [B, P, ND + ND^2] (RENI.py:23-28)."""
    B, P = D.shape[0], D.shape[1]
    gram = (Z @ Z.transpose(1, 2)).reshape(B, 1, -1).expand(B, P, -1)
    return torch.cat((D @ Z.transpose(1, 2), gram), 2)


def SO2InvariantRepresentation(Z, D):
    """[D_xz Z_xz^T | vec(Z_xz Z_xz^T) | |d_xz| | Z_y | d_y] -> [B, P, 2ND + ND^2 + 2] (RENI.py:31-53)."""
    B, P = D.shape[0], D.shape[1]
    z_xz, d_xz = Z[:, :, [0, 2]], D[:, :, [0, 2]]
    gram = (z_xz @ z_xz.transpose(1, 2)).reshape(B, 1, -1).expand(B, P, -1)
    radius = torch.sqrt(D[:, :, 0] ** 2 + D[:, :, 2] ** 2).unsqueeze(2)
    z_y = Z[:, :, 1].unsqueeze(1).expand(B, P, -1)
    return torch.cat((d_xz @ z_xz.transpose(1, 2), gram, radius, z_y, D[:, :, 1:2]), 2)


def NoInvariance(Z, D):
    """[D Z^T | vec(Z)] -> [B, P, 4ND] (RENI.py:56-60)."""
    B, P = D.shape[0], D.shape[1]
    return torch.cat((D @ Z.transpose(1, 2), Z.reshape(B, 1, -1).expand(B, P, -1)), 2)


# ---- FiLM conditioning: (Siren_Input, Mapping_Input) pairs (RENI.py:407-452) -------------------


def SO3InvariantRepresentationFiLM(Z, D):
    """(D Z^T [B,P,ND], vec(Z Z^T) per pixel [B,P,ND^2]) (RENI.py:407-415)."""
    B, P = D.shape[0], D.shape[1]
    gram = (Z @ Z.transpose(1, 2)).reshape(B, 1, -1).expand(B, P, -1)
    return D @ Z.transpose(1, 2), gram


def SO2InvariantRepresentationFiLM(Z, D):
    """([|d_xz| | d_y | D_xz Z_xz^T] [B,P,2+ND], [vec(Z_xz Z_xz^T) | Z_y] per pixel [B,P,ND^2+ND]) (RENI.py:418-447)."""
    B, P = D.shape[0], D.shape[1]
    z_xz, d_xz = Z[:, :, [0, 2]], D[:, :, [0, 2]]
    gram = (z_xz @ z_xz.transpose(1, 2)).reshape(B, 1, -1).expand(B, P, -1)
    radius = torch.sqrt(D[:, :, 0] ** 2 + D[:, :, 2] ** 2).unsqueeze(2)
    z_y = Z[:, :, 1].unsqueeze(1).expand(B, P, -1)
    return torch.cat((radius, D[:, :, 1:2], d_xz @ z_xz.transpose(1, 2)), 2), torch.cat((gram, z_y), 2)


def NoInvarianceFiLM(Z, D):
    """(D Z^T, vec(Z) per pixel) (RENI.py:449-452)."""
    B, P = D.shape[0], D.shape[1]
    return D @ Z.transpose(1, 2), Z.reshape(B, 1, -1).expand(B, P, -1)
