"""UV / angle / depth map rendering of a UV-parameterised scene mesh."""
from .rasterizer import Mesh, box_room_mesh, load_obj, render_maps, render_trajectory, scaled_intrinsics  # noqa: F401
