"""Drop-in mirror of the reference's module tree for the rendering hot path.

    reference module            ours
    src.utils.Renderer     ->   myslam_amd.src.utils.Renderer   (Renderer)
    src.common             ->   myslam_amd.src.common           (get_samples, get_rays, normalize_3d_coordinate, ...)
    src.networks.decoders  ->   myslam_amd.src.networks.decoders (Decoders)
    src.config.get_model   ->   myslam_amd.src.config.get_model

Same names, argument order, return shapes and error behaviour (plain exceptions) as the reference, so
reference src/Mapper.py and src/Tracker.py call them unchanged.
"""
