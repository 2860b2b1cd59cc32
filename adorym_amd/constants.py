"""Constants of the reference that the hot path depends on."""
# adorym/constants.py:90 -- deliberately the reference's truncated value (propagate.py:20 star-imports it)
PI = 3.14159265359
