"""Column indices of the 11-d box state used throughout the decoder
(reference: projects/mmdet3d_plugin/core/box3d.py:1-3 -- an index convention, kept identical)."""
X, Y, Z, W, L, H, SIN_YAW, COS_YAW, VX, VY, VZ = range(11)  # un-decoded box: xyz, log-size, yaw as (sin, cos), velocity
CNS, YNS = 0, 1  # centerness / yawness columns of the quality head
YAW = 6  # decoded boxes carry the yaw angle here
