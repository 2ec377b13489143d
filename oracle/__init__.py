"""CPU oracle — TEST INFRASTRUCTURE ONLY (see oracle/oracle.py)."""
