"""CPU ORACLE -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product package (pointcloudregistration_amd) never does.
"""
