"""Volume rendering of the tri-plane avatar (mirror of the reference package)."""
