from .Calib import lidar2Img, lidar2P2, p22Lidar  # noqa: F401
