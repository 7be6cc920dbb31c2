"""UV / angle / depth map rendering of a UV-parameterised scene mesh."""
from .rasterizer import (Mesh, box_room_mesh, build_mipmaps, load_obj, project_points, render_maps,  # noqa: F401
                         render_textured, render_trajectory, sample_mipmapped, save_obj, scaled_intrinsics)
