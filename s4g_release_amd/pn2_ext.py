"""Drop-in for the reference's pybind11 module ``pn2_ext``.

Exposes the seven names registered at
``pointnet2_utils/csrc/main.cpp:7-13`` with the same positional signatures, so
reference-shaped code (``from . import pn2_ext`` in ``functions.py:2``) binds
to the MI355X kernels unchanged.  See INTEGRATION.md.
"""
from .functions import (_ball_query as ball_query,  # noqa: F401
                        _farthest_point_sample as farthest_point_sample,
                        _group_points_backward as group_points_backward,
                        _group_points_forward as group_points_forward,
                        _interpolate_backward as interpolate_backward,
                        _interpolate_forward as interpolate_forward,
                        _point_search as point_search)

__all__ = ["ball_query", "group_points_forward", "group_points_backward", "farthest_point_sample",
           "point_search", "interpolate_forward", "interpolate_backward"]
