"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; nothing under deeprl_signal_control_amd/ does.  See DESIGN.md
"Oracle"."""
