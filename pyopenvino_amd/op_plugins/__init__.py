"""One module per IR layer type; the module name is the layer type (reference inference_engine.py:28-43)."""
