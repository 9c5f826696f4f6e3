"""Frechet distance between two sets of feature vectors (the distance part of src/fid.py:95-163) and a labelled PROXY of
the reference's FID.

The reference's FID extracts pool features with a pretrained Inception-v3 (src/fid.py:32-93); those weights cannot be
obtained offline, so ``fid_proxy`` uses the DISCRIMINATOR's trunk as the feature extractor (the activation in front of its
head, spatially averaged to (N, C) exactly as the reference averages Inception's Mixed_7c map, eval-mode BatchNorm) and
says so in its name.  The statistics and the distance are the reference's: mu = mean, sigma = np.cov(rowvar=False);
d^2 = |mu1 - mu2|^2 + Tr(C1 + C2 - 2 (C1 C2)^(1/2)) with the matrix square root from scipy and the usual remedies for a
(near-)singular product (add eps to the diagonals) and for a small imaginary part from round-off.
"""
from __future__ import annotations

import warnings

import numpy as np
import torch
from scipy import linalg


def activation_statistics(act):
    """(mu, sigma) of an (N, F) array of feature vectors (src/fid.py:106-109)."""
    act = np.asarray(act, dtype=np.float64)
    return act.mean(axis=0), np.cov(act, rowvar=False)


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """Squared Frechet distance between N(mu1, sigma1) and N(mu2, sigma2) (src/fid.py:112-163)."""
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    if mu1.shape != mu2.shape or sigma1.shape != sigma2.shape:
        raise ValueError("the two sets of statistics have different dimensions")
    delta = mu1 - mu2
    root, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(root).all():          # singular product: regularise both covariances
        warnings.warn("frechet_distance: singular covariance product; adding %g to the diagonals" % eps)
        jitter = np.eye(sigma1.shape[0]) * eps
        root = linalg.sqrtm((sigma1 + jitter).dot(sigma2 + jitter))
    if np.iscomplexobj(root):                # round-off can leave a tiny imaginary part
        if not np.allclose(np.diagonal(root).imag, 0, atol=1e-3):
            raise ValueError("frechet_distance: imaginary component %g" % np.max(np.abs(root.imag)))
        root = root.real
    return float(delta.dot(delta) + np.trace(sigma1) + np.trace(sigma2) - 2.0 * np.trace(root))


@torch.no_grad()
def discriminator_features(discriminator, images, batch_size=64):
    """(N, C) features: the discriminator trunk's last activation (eval-mode BatchNorm), averaged over its 4x4 map.
    ``images``: (N, 3, S, S) float tensor in [-1, 1] (the GAN's own normalisation)."""
    was_training = discriminator.training
    discriminator.eval()
    dev = next(discriminator.parameters()).device
    feats = []
    try:
        for i in range(0, images.shape[0], batch_size):
            f = discriminator(images[i:i + batch_size].to(dev).float(), feature_matching=True)
            feats.append(f.mean(dim=(2, 3)).cpu().numpy())
    finally:
        discriminator.train(was_training)
    return np.concatenate(feats, axis=0)


def fid_proxy(discriminator, images1, images2, batch_size=64):
    """Frechet distance between the discriminator-trunk features of two image sets.  NOT the Inception FID of
    src/fid.py:217-232 (no Inception weights offline): comparable only between runs that use the same discriminator."""
    m1, s1 = activation_statistics(discriminator_features(discriminator, images1, batch_size))
    m2, s2 = activation_statistics(discriminator_features(discriminator, images2, batch_size))
    return frechet_distance(m1, s1, m2, s2)


# ------------------------------------------------------------------------------------------------------------------
# The rest of the reference's FID procedure around the feature extractor (src/fid.py:166-232, :312-330).  The extractor
# itself (Inception-v3 Mixed_7c pool features, :33-94) needs pretrained weights that are not obtainable offline, so it
# is a parameter here: any callable (N, 3, 299, 299) float tensor in [0, 1] -> (N, F) array.
# ------------------------------------------------------------------------------------------------------------------
def preprocess_image(im):
    """src/fid.py:166-190: (H, W, 3) uint8 or float32 in [0, 1] -> (3, 299, 299) float32 in [0, 1].  The reference
    resizes with cv2.resize(im, (299, 299)) (bilinear, half-pixel centres, no anti-aliasing); cv2 is absent here, the
    same sampling rule is applied with torch (parity with cv2 unpinned: no cv2 to compare with)."""
    im = np.asarray(im)
    if im.ndim != 3 or im.shape[2] != 3:
        raise ValueError("preprocess_image expects an (H, W, 3) image")
    if im.dtype == np.uint8:
        im = im.astype(np.float32) / 255
    t = torch.from_numpy(np.ascontiguousarray(im, dtype=np.float32)).permute(2, 0, 1)[None]
    t = torch.nn.functional.interpolate(t, size=(299, 299), mode="bilinear", align_corners=False, antialias=False)[0]
    if float(t.max()) > 1.0 or float(t.min()) < 0.0:
        raise ValueError("preprocess_image: values outside [0, 1]")
    return t.contiguous()


def preprocess_images(images, use_multiprocessing=False):
    """src/fid.py:193-214: (N, H, W, 3) -> (N, 3, 299, 299) float32 in [0, 1] (use_multiprocessing is accepted for
    signature compatibility; the resize is a single batched op here)."""
    out = torch.stack([preprocess_image(im) for im in images], dim=0)
    assert out.shape == (len(images), 3, 299, 299) and out.dtype == torch.float32
    return out


def calculate_fid(images1, images2, feature_extractor, batch_size=2, use_multiprocessing=False):
    """src/fid.py:217-232 with the feature extractor as a parameter: images (N, H, W, 3) uint8 / float in [0, 1]."""
    def feats(images):
        x = preprocess_images(images, use_multiprocessing)
        return np.concatenate([np.asarray(feature_extractor(x[i:i + batch_size]), dtype=np.float64)
                               for i in range(0, x.shape[0], batch_size)], axis=0)
    m1, s1 = activation_statistics(feats(images1))
    m2, s2 = activation_statistics(feats(images2))
    return frechet_distance(m1, s1, m2, s2)


def inception_feature_extractor(weights, device="cuda:0"):
    """The reference's feature extractor (src/fid.py:33-94: torchvision inception_v3 up to Mixed_7c, spatially averaged) on
    the HIP kernels, as the ``feature_extractor`` argument of calculate_fid / fid_protocol.  ``weights``: a torchvision
    ``inception_v3`` state_dict (or the path of a file holding one) -- pretrained weights are not obtainable offline, they
    are an input here.  Returns a callable (N, 3, 299, 299) float tensor in [0, 1] -> (N, 2048) float32 array."""
    from .inception import InceptionV3
    net = InceptionV3()
    if isinstance(weights, (str, bytes)):
        weights = torch.load(weights, map_location="cpu")
    net.load_state_dict(weights)
    net = net.to(device).eval()

    def extract(x01):
        return net.features(x01.to(device)).cpu().numpy()
    return extract


def fid_protocol(generate_fake, real_images, feature_extractor, iterations=5, batch_size=2):
    """The reference's reporting protocol (src/fid.py:312-330): `iterations` (= 5) independent generations of the fake
    set against the same real set, FID of each, reported as mean +- std.  generate_fake() -> (N, H, W, 3) images."""
    values = [calculate_fid(real_images, generate_fake(), feature_extractor, batch_size) for _ in range(iterations)]
    return {"fid_values": values, "mean": float(np.mean(values)), "std": float(np.std(values))}
