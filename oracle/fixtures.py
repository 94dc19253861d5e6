"""TEST INFRASTRUCTURE -- shared deterministic inputs (thin re-export of the product's
synthetic-input helpers so that the reference run and the tests build identical tensors)."""
from selfreconcode_amd.synthetic import (  # noqa: F401
    det_array, det_tensor, det_params, det_normal, sphere_sdf_params, synthetic_joints, synthetic_lbs_volume,
    SDF_SPEC, DEF_SPEC, REND_SPEC, LBS_BMIN, LBS_BMAX, SMPL_PARENTS, cube_sphere, icosphere,
)
