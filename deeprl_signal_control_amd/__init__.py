"""MI355X-native traffic-signal RL hot path (batched HIP microsimulator +
fused per-agent A2C nets) behind the env/model duck-types of
cts198859/deeprl_signal_control.  See DESIGN.md."""
__version__ = '0.1.0'
