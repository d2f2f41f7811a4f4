"""ORACLE plugins -- test infrastructure only.  One module per IR layer type with the reference's
compute(node, inputs, kernel_type, debug) signature, numeric bodies from oracle/ops.py (numpy, CPU).
Loaded by tests / the cpu_baseline leg of bench.py with IECore(plugin_package='oracle.op_plugins')."""
