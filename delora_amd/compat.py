"""Import-name compatibility with the reference, whose packages are top-level (``import deploy.trainer``,
``utility.projection``, ``losses.icp_losses``, ``models.model``, ``preprocessing.normal_computation``,
``data.dataset``, ``ros_utils``; reference setup.py installs ``src/`` as the package root).  Importing this module registers
the delora_amd sub-packages under those names."""
import importlib
import sys

_NAMES = ("deploy", "utility", "losses", "models", "preprocessing", "data", "ros_utils")
_SUBMODULES = {
    "deploy": ("deployer", "trainer", "tester", "step_geometry"), "utility": ("projection", "poses"), "losses": ("icp_losses",),
    "models": ("model", "model_parts", "resnet_modified"), "preprocessing": ("normal_computation", "preprocesser"),
    "data": ("dataset", "synthetic", "feed", "kitti_scans"),
    "ros_utils": ("odometry",),            # the ROS-free inference core only: the ROS node itself is out of scope and not built
}

for _name in _NAMES:
    _pkg = importlib.import_module("delora_amd." + _name)
    sys.modules.setdefault(_name, _pkg)
    for _sub in _SUBMODULES[_name]:
        _mod = importlib.import_module(f"delora_amd.{_name}.{_sub}")
        sys.modules.setdefault(f"{_name}.{_sub}", _mod)
