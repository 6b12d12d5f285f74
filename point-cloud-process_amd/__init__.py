"""MI355X-native registration hot path (ICP + nearest neighbour + voxel filter).

Drop-in for the reference's Registration/main.py, Registration/icp_template.py,
Kdtree_Octree/lesson2 and Pca_and_Voxel_filter/voxel_filter.py call surfaces,
backed by hand-written HIP kernels for gfx950 behind the C ABI in include/pcr.h.

The directory name contains a hyphen, so import it with
``importlib.import_module("point-cloud-process_amd")`` or through the
top-level alias module ``pcp_amd``.
"""
from . import _lib  # noqa: F401
from . import synthetic  # noqa: F401
from .device import Context, DeviceCloud, TargetIndex, default_context, icp_device  # noqa: F401
from .registration import (  # noqa: F401
    ICP,
    coarse_to_fine_icp,
    KDTreeFlann,
    PointCloud,
    copysign,
    find_associations,
    homo2tq,
    icp_point2point,
    procrustes_transformation,
    read_bin_velodyne,
    read_oxford_bin,
    read_velodyne_bin,
    rotmat2quaternion,
    write_reg_result,
)

from .voxel_filter import voxel_filter, voxel_filter_device, voxel_keys  # noqa: F401
from .result_set import DistIndex, KNNResultSet, RadiusNNResultSet  # noqa: F401
from .kdtree import kdtree_construction, kdtree_knn_search, kdtree_radius_search, knn_search_batch, radius_search_batch  # noqa: F401
from .octree import octree_construction, octree_knn_search, octree_radius_search, octree_radius_search_fast  # noqa: F401
from .iss import iss_keypoints  # noqa: F401
from .pca_normal import PCA, estimate_normals  # noqa: F401
from .global_registration import (  # noqa: F401
    Feature,
    RegistrationResult,
    compute_fpfh_feature,
    estimate_normals_hybrid,
    execute_global_registration,
    find_matchings,
    prepare_dataset,
    preprocess_point_cloud,
    ransac_init,
    registration_ransac_based_on_feature_matching,
    voxel_down_sample,
)
from .dbscan import DBSCAN  # noqa: F401
from .batch import register_batch, shard_range  # noqa: F401
from . import evaluate  # noqa: F401
from .evaluate import evaluate_rt, get_P_diff, is_registration_successful  # noqa: F401
from .drivers import read_pair_list, run_registration  # noqa: F401

__version__ = "0.1.0"
