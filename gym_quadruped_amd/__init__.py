"""MI355X-native batched drop-in for gym-quadruped's ``QuadrupedEnv.step`` hot path."""
__version__ = '0.1.0'
