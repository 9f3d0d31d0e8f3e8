/*
 * bdsf_list.h -- the material plugin list (X-macro), same role, names and ORDER as the
 * reference's src/bdsf_list.h:1-14. A .scn material names its scattering functions
 * (`bdsfs a, b, ...`) and its direction sampler (`dir_func f`) by these identifiers.
 *
 * The reference expands the list into prototypes, name tables and function-pointer tables
 * (src/bdsf.h:4-52). Here it is expanded into
 *   - integer IDs                       include/drt_hip.h   (DRT_BDSF_<name>, DRT_DIRF_<name>)
 *   - name tables for the .scn parser   daily-ray-trace_amd/host/drt_scene.c   (bdsf_name_list, dir_func_name_list)
 *   - a device-side `switch`            daily-ray-trace_amd/csrc/drt_kernels.h  (bdsf_at_wavelength, sample_direction)
 * Adding a scattering function = one BDSF() line here + one device function + one oracle function.
 *
 * Define BDSF(name) and DIRF(name) before including; no include guard on purpose.
 */

/* scattering functions: out = f(point, incoming) over all wavelengths */
BDSF(bp_diffuse_bdsf)                  /* Lambert term of the Blinn-Phong plastic  */
BDSF(bp_glossy_bdsf)                   /* Blinn-Phong lobe, exponent = shininess   */
BDSF(mirror_bdsf)                      /* RGB-tinted perfect mirror                */
BDSF(fs_conductor_bdsf)                /* smooth conductor, per-wavelength Fresnel */
BDSF(fs_dielectric_reflectance_bdsf)   /* smooth dielectric, reflected part        */
BDSF(fs_dielectric_transmittance_bdsf) /* smooth dielectric, refracted part        */
BDSF(ct_conductor_bdsf)                /* Cook-Torrance GGX rough conductor        */

/* direction samplers: (incoming direction, reciprocal pdf) = g(point) */
DIRF(cos_weighted_sample_hemisphere)
DIRF(uniform_sample_hemisphere)
DIRF(sample_specular_direction)
DIRF(sample_transmit_direction)
DIRF(sample_reflect_or_transmit_direction)
DIRF(sample_ct_direction)
