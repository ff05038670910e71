"""paddlerobotics_amd -- MI355X-native batched A1 quadruped simulator behind the
ETGRL env.step()/reset() Gym surface."""
__version__ = "0.1.0"
