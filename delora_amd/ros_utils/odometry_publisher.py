"""ROS node around ``ScanToScanOdometry``: subscribes to a PointCloud2 topic, publishes ``/delora/odometry`` and, if asked,
the integrated pose on TF (reference src/ros_utils/odometry_publisher.py:27-197, odometry_integrator.py:26-105).
Needs a ROS 1 Python environment (rospy, ros_numpy, tf2_ros, geometry/nav/sensor message packages); importing this module
without one raises ImportError with that message.  All computation is in ``odometry.py``."""
import numpy as np

try:
    import geometry_msgs.msg
    import nav_msgs.msg
    import ros_numpy
    import rospy
    import sensor_msgs.msg
    import tf2_ros
except ImportError as exc:                                              # pragma: no cover - no ROS in the build image
    raise ImportError("delora_amd.ros_utils.odometry_publisher needs a ROS 1 Python environment (rospy, ros_numpy, tf2_ros, "
                      "geometry_msgs, nav_msgs, sensor_msgs); the ROS-free core is delora_amd.ros_utils.odometry") from exc

from . import odometry


class OdometryIntegrator(object):                                       # pragma: no cover
    """TF side of the integrated pose: world -> lidar from ``T_0_t``, lidar -> delora_odom fixed (odometry_integrator.py)."""

    def __init__(self, config):
        self.lidar_frame = config["lidar_frame"]
        self.odom_frame, self.world_frame = "delora_odom", "world"
        self.tf_broadcaster = tf2_ros.TransformBroadcaster()
        self.T_pc_odom = self._identity(self.lidar_frame, self.odom_frame)
        self.T_world_pc = self._identity(self.world_frame, self.lidar_frame)
        self.tf_broadcaster.sendTransform(self.T_world_pc)
        self.tf_broadcaster.sendTransform(self.T_pc_odom)

    @staticmethod
    def _identity(parent, child):
        t = geometry_msgs.msg.TransformStamped()
        t.header.frame_id, t.child_frame_id = parent, child
        t.transform.rotation.w = 1.0
        return t

    def publish(self, header, global_translation, global_quaternion):
        tr, ro = self.T_world_pc.transform.translation, self.T_world_pc.transform.rotation
        tr.x, tr.y, tr.z = (float(v) for v in global_translation)
        ro.x, ro.y, ro.z, ro.w = (float(v) for v in global_quaternion)
        self.T_world_pc.header.stamp = header.stamp
        self.tf_broadcaster.sendTransform(self.T_world_pc)
        self.T_pc_odom.header.stamp = header.stamp
        self.tf_broadcaster.sendTransform(self.T_pc_odom)


class OdometryPublisher(object):                                        # pragma: no cover
    def __init__(self, config):
        self.config = config
        self.lidar_topic, self.lidar_frame = config["lidar_topic"], config["lidar_frame"]
        self.core = odometry.ScanToScanOdometry(config)
        self.odometry_publisher = rospy.Publisher("/delora/odometry", nav_msgs.msg.Odometry, queue_size=10)
        rospy.init_node("LiDAR_odometry_publisher", anonymous=True)
        self.tf = OdometryIntegrator(config) if config["integrate_odometry"] else None
        self.odometry_ros = nav_msgs.msg.Odometry()

    def subscriber_callback(self, data):
        cloud = ros_numpy.numpify(data)
        scan = np.stack((cloud["x"].view(np.float32).reshape(-1), cloud["y"].view(np.float32).reshape(-1),
                         cloud["z"].view(np.float32).reshape(-1)), axis=0)[None]
        out = self.core.push(scan)
        if out is None:
            return
        pose = self.odometry_ros.pose.pose
        pose.position.x, pose.position.y, pose.position.z = (float(v) for v in out["translation"])
        o = pose.orientation
        o.x, o.y, o.z, o.w = (float(v) for v in out["quaternion"])
        self.odometry_ros.header = data.header
        self.odometry_ros.header.frame_id = self.lidar_frame
        self.odometry_publisher.publish(self.odometry_ros)
        if self.tf is not None:
            self.tf.publish(data.header, out["global_translation"], out["global_quaternion"])

    def publish_odometry(self):
        rospy.Subscriber(self.lidar_topic, sensor_msgs.msg.PointCloud2, self.subscriber_callback)
        rospy.spin()
