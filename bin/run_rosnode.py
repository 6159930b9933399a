#!/usr/bin/env python3
"""Inference node with the reference's command line (bin/run_rosnode.py:16-26):

    python bin/run_rosnode.py --checkpoint FILE --dataset kitti --lidar_topic /velodyne_points --lidar_frame velodyne

subscribes to the PointCloud2 topic and publishes the scan-to-scan odometry (needs a ROS 1 Python environment; the
computation itself is delora_amd/ros_utils/odometry.py and runs without ROS).
"""
import os
import sys

import click

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import delora_amd.compat  # noqa: E402,F401
import delora_amd.config  # noqa: E402


@click.command()
@click.option("--checkpoint", prompt="Path to the saved model you want to test")
@click.option("--dataset", prompt="On which dataset configuration do you want to get predictions? [kitti, darpa, ....]. Does not "
                                  "need to be one of those, but the sensor paramaters are looked up in the config_datasets.yaml.")
@click.option("--lidar_topic", prompt="Topic of the published LiDAR pointcloud2 messages.")
@click.option("--lidar_frame", prompt="LiDAR frame in TF tree.")
@click.option("--integrate_odometry", help="Whether the published odometry should be integrated in the TF tree.", default=True)
def config(checkpoint, dataset, lidar_topic, lidar_frame, integrate_odometry):
    cfg = delora_amd.config.rosnode_config(checkpoint, dataset, lidar_topic, lidar_frame, integrate_odometry)
    print("----------------------------------")
    print("Configuration for this run: ")
    print(cfg)
    print("----------------------------------")
    return cfg


if __name__ == "__main__":
    cfg = config(standalone_mode=False)
    import delora_amd.ros_utils.odometry_publisher as odometry_publisher
    publisher = odometry_publisher.OdometryPublisher(config=cfg)
    publisher.publish_odometry()
