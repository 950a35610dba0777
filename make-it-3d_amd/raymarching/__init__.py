from .raymarching import *  # noqa: F401,F403  (same star-export as the reference's raymarching/__init__.py:1)
from .raymarching import (composite_rays, composite_rays_train, composite_sdf_rays, composite_sdf_rays_train,  # noqa: F401
                          march_rays, march_rays_train, morton3D, morton3D_invert, near_far_from_aabb, packbits,
                          sph_from_ray)
