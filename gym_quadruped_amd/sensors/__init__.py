from .base_sensor import Sensor  # noqa: F401
from .heightmap import HeightMap  # noqa: F401
from .imu import IMU  # noqa: F401
