"""Lightning / Hydra-free drivers around the three plugins: config loading
from the reference's `confs/` layout, Lightning-compatible checkpoint I/O, and the `animate.py`
equivalent.  They add no compute of their own: every frame goes through the same plugin calls
`DNeRFModel.render_image_fast` makes."""
