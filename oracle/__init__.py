"""CPU oracle package - TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this.  The product path (gym_quadruped_amd/) never does."""
