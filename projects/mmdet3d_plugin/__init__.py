"""MI355X-native drop-in for the hot path of HiP-AD's ``projects.mmdet3d_plugin``.

Same package path, module names and public symbols as the reference for the rows of
SURVEY.md section 8 (ops, DeformableFeatureAggregation, decoder blocks); everything
underneath is new code on top of ``hip-ad_amd`` (HIP kernels behind a C ABI).
"""
