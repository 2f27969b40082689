"""Alternate keyword spellings accepted next to the reference's own (btk20cpp/_signatures.py, generated from the .i files): the
names earlier versions of this package gave the same parameters, kept so that callers written against them keep working, and the
legacy camelCase aliases (ENABLE_LEGACY_BTK_API) the binding does not define itself.  (name, default) per parameter; None = required."""

METHOD_KWARGS = {
    'blocking_matrix_output': [('out_chan_no', 0)],
    'calc_array_manifold_vectors_2': [('samplerate', None), ('delays_t', None), ('delays_j', None)],
    'calc_blocking_matrix1': [('samplerate', None), ('delays_t', None)],
    'calc_gsc_weights': [('samplerate', None), ('delays_t', None)],
    'calc_gsc_weights_2': [('samplerate', None), ('delays_t', None), ('delays_i', None)],
    'calc_gsc_weights_n': [('samplerate', None), ('delays_t', None), ('delays_is', None), ('NC', 2)],
    'calc_mvdr_weights': [('samplerate', None), ('dthreshold', 1e-08), ('calc_inverse_matrix', True)],
    'divide_all_nondiagonal_elements': [('mu', None)],
    'divide_nondiagonal_elements': [('fbin_no', None), ('mu', None)],
    'estimate_filter': [('start_frame_no', 0), ('end_frame_no', -1)],
    'getSpecMatrix': [('idx', None)],
    'getWeights': [('fbin_no', None)],
    'get_weights': [('fbin_no', None)],
    'matrix_f': [('idx', None)],
    'mvdr_weights': [('fbin_no', None)],
    'noise_spatial_spectral_matrix': [('fbin_no', None)],
    'normalize_weight': [('flag', None)],
    'print_objective_func': [('subbandX', None)],
    'setBeamformer': [('bf', None)],
    'setChannel': [('chan', None)],
    'set_active_weights_f': [('fbin_no', None), ('packed_weight', None)],
    'set_all_diagonal_loading': [('diagonal_weight', None)],
    'set_beamformer': [('bf', None)],
    'set_channel': [('chan', None)],
    'set_diagonal_looading': [('fbin_no', None), ('diagonal_weight', None)],
    'set_input': [('samples', None)],
    'set_noise_spatial_spectral_matrix': [('fbin_no', None), ('Rnn', None)],
    'set_precision_matrix': [('fbin_no', None), ('Pz', None)],
    'set_quiescent_weights_f': [('fbin_no', None), ('src_wq', None)],
    'set_samples': [('samples', None)],
    'snapshot': [('fbin_no', None)],
    'snapshot_array_f': [('fbin_no', None)],
    'update_active_weight_vecotrs': [('flag', None)],
}

# where a class names the parameters of a method differently
CLASS_METHOD_KWARGS = {
    'SingleChannelWPEDereverberationFeaturePtr': {'print_objective_func': [('subband_no', None)]},
    'SnapShotArrayPtr': {'set_samples': [('samp', None), ('chan_no', None)]},
    'SpectralMatrixArrayPtr': {'set_samples': [('samp', None), ('chan_no', None)]},
}

CTOR_KWARGS = {
    'LefkimmiatisPostFilterPtr': [('output', None), ('fftlen', None), ('min_sv', 1e-08), ('fbin_no1', 0), ('alpha', 0.6), ('type', 2), ('min_frames', 0), ('threshold', 0.99), ('nm', 'LefkimmiatisPostFilterPtr')],
    'PyVectorComplexFeatureStreamPtr': [('obj', None), ('name', 'PyVectorComplexFeatureStream')],
    'PyVectorFloatFeatureStreamPtr': [('obj', None), ('name', 'PyVectorFloatFeatureStream')],
    'SnapShotArrayPtr': [('fftlen', None), ('chan_num', None)],
}

ALIASES = {
    'LefkimmiatisPostFilterPtr': {'calcInverseNoiseSpatialSpectralMatrix': 'calc_inverse_noise_spatial_spectral_matrix'},
    'McCowanPostFilterPtr': {'getNoiseSpatialSpectralMatrix': 'noise_spatial_spectral_matrix', 'setNoiseSpatialSpectralMatrix': 'set_noise_spatial_spectral_matrix', 'setDiffuseNoiseModel': 'set_diffuse_noise_model', 'setAllLevelsOfDiagonalLoading': 'set_all_diagonal_loading', 'setLevelOfDiagonalLoading': 'set_diagonal_looading', 'divideAllNonDiagonalElements': 'divide_all_nondiagonal_elements', 'divideNonDiagonalElements': 'divide_nondiagonal_elements'},
    'MultiChannelWPEDereverberationPtr': {'setInput': 'set_input', 'nextSpeaker': 'next_speaker'},
    'SingleChannelWPEDereverberationFeaturePtr': {'nextSpeaker': 'next_speaker'},
    'SnapShotArrayPtr': {'fftlen': 'fftLen', 'chan_num': 'nChan'},
    'SubbandDSPtr': {'calcArrayManifoldVectors': 'calc_array_manifold_vectors'},
    'SubbandGSCPtr': {'calcGSCWeights': 'calc_gsc_weights', 'setActiveWeights_f': 'set_active_weights_f', 'getBlockingMatrix': 'blocking_matrix'},
    'SubbandGSCRLSPtr': {'initPrecisionMatrix': 'init_precision_matrix', 'setPrecisionMatrix': 'set_precision_matrix', 'updateActiveWeightVecotrs': 'update_active_weight_vecotrs', 'setQuadraticConstraint': 'set_quadratic_constraint'},
    'SubbandMVDRGSCPtr': {'upgradeBlockingMatrix': 'upgrade_blocking_matrix', 'blockingMatrixOutput': 'blocking_matrix_output'},
    'SubbandMVDRPtr': {'setDiffuseNoiseModel': 'set_diffuse_noise_model', 'setAllLevelsOfDiagonalLoading': 'set_all_diagonal_loading', 'calcMVDRWeights': 'calc_mvdr_weights', 'getMVDRWeights': 'mvdr_weights'},
    'ZelinskiPostFilterPtr': {'getPostFilterWeights': 'postfilter_weights'},
}
