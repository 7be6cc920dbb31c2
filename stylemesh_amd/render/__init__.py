"""UV / angle / depth map rendering of a UV-parameterised scene mesh."""
from .rasterizer import (Mesh, box_room_mesh, build_mipmaps, load_obj, render_maps, render_textured,  # noqa: F401
                         render_trajectory, sample_mipmapped, scaled_intrinsics)
