"""CPU oracle (test infrastructure only).  See oracle/oracle_np.py and oracle/oracle.c.

Nothing in the product package may import this package."""
