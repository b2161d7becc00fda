// Reader of the PhaseSeries hand-off file (INTEGRATION.md "Archive hand-off"; written by dspsr_amd/pipeline.py
// write_phase_series) for the DSPSR side: fills a host dsp::PhaseSeries with exactly the members dsp::Archiver::set
// reads (Signal/Pulsar/Archiver.C:430-893), so that `archiver->unload (&phase_series)` writes the archive.
// Two layers: read_phase_series_file() is plain C++ (no DSPSR types; tests/test_phase_series_io.py runs it), and
// load_phase_series() maps the result onto the public API of dsp::PhaseSeries / dsp::Observation
// (Signal/Pulsar/dsp/PhaseSeries.h:28-205, Kernel/Classes/dsp/Observation.h:50-205) -- compiled only where
// "dsp/PhaseSeries.h" has been included first.
#ifndef DSPSR_AMD_PHASE_SERIES_IO_H
#define DSPSR_AMD_PHASE_SERIES_IO_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace HIP
{
  struct PhaseSeriesFile
  {
    std::map<std::string, std::string> header;   // "KEY value" lines of the 4096-byte ASCII header
    unsigned nchan, npol, ndim, nbin;
    std::vector<uint32_t> hits;                  // [nbin]
    std::vector<float> sums;                     // [nchan][npol][nbin][ndim], un-normalised (Archiver divides by scale*hits)

    double number (const std::string& key) const
    {
      std::map<std::string, std::string>::const_iterator it = header.find (key);
      if (it == header.end ()) throw std::runtime_error ("PhaseSeries file: key " + key + " missing");
      return atof (it->second.c_str ());
    }
    std::string text (const std::string& key) const
    {
      std::map<std::string, std::string>::const_iterator it = header.find (key);
      return it == header.end () ? std::string () : it->second;
    }
  };

  inline PhaseSeriesFile read_phase_series_file (const char* filename)
  {
    PhaseSeriesFile f;
    FILE* fp = fopen (filename, "rb");
    if (!fp) throw std::runtime_error (std::string ("PhaseSeries file: cannot open ") + filename);
    std::vector<char> hdr (4096 + 1, 0);
    if (fread (&hdr[0], 1, 4096, fp) != 4096) { fclose (fp); throw std::runtime_error ("PhaseSeries file: truncated header"); }
    for (char* line = strtok (&hdr[0], "\n"); line; line = strtok (0, "\n"))
    {
      char key[64], value[256];
      if (sscanf (line, "%63s %255[^\n]", key, value) == 2)
      {
        std::string v (value);
        while (!v.empty () && (v[v.size () - 1] == ' ' || v[v.size () - 1] == '\r')) v.erase (v.size () - 1);
        f.header[key] = v;
      }
    }
    if (f.text ("HDR_MAGIC") != "DSPSR_AMD_PHASESERIES")
    { fclose (fp); throw std::runtime_error (std::string (filename) + " is not a DSPSR_AMD_PHASESERIES file"); }
    f.nchan = unsigned (f.number ("NCHAN")); f.npol = unsigned (f.number ("NPOL"));
    f.ndim = unsigned (f.number ("NDIM")); f.nbin = unsigned (f.number ("NBIN"));
    f.hits.resize (f.nbin);
    f.sums.resize (size_t (f.nchan) * f.npol * f.nbin * f.ndim);
    const bool ok = fread (&f.hits[0], sizeof (uint32_t), f.nbin, fp) == f.nbin &&
                    fread (&f.sums[0], sizeof (float), f.sums.size (), fp) == f.sums.size ();
    fclose (fp);
    if (!ok) throw std::runtime_error (std::string (filename) + " is truncated");
    return f;                                     // (little-endian file, little-endian hosts only)
  }

#ifdef __PhaseSeries_h
  //! Fill a host dsp::PhaseSeries from the file: what Fold::get_result() would have handed to the Archiver
  inline void load_phase_series (const char* filename, dsp::PhaseSeries* out)
  {
    const PhaseSeriesFile f = read_phase_series_file (filename);
    const std::string state = f.text ("STATE");
    out->set_centre_frequency (f.number ("FREQ"));
    out->set_bandwidth (f.number ("BW"));
    out->set_nchan (f.nchan);
    out->set_npol (f.npol);
    out->set_ndim (f.ndim);
    out->set_state (state == "Stokes" ? Signal::Stokes : state == "Coherence" ? Signal::Coherence :
                    state == "PPQQ" ? Signal::PPQQ : Signal::Intensity);
    out->set_dispersion_measure (f.number ("DM"));
    out->set_scale (f.number ("SCALE"));             // Archiver.C:842: amps = sum / (scale * hits)
    const double sec = f.number ("MJD_SEC") + f.number ("OBS_OFFSET_SECONDS");
    out->set_start_time (MJD (int (f.number ("MJD_DAY")), int (sec), sec - int (sec)));
    out->resize (f.nbin);                            // PhaseSeries.C:83-110: the profile rows and hits[nbin]
    out->zero ();
    const double period = f.number ("FOLDING_PERIOD");
    if (period > 0) out->set_folding_period (period);
    out->set_reference_phase (f.number ("REFERENCE_PHASE"));
    out->increment_integration_length (f.number ("INTEGRATION_LENGTH"));
    out->set_ndat_expected (uint64_t (f.number ("NDAT_TOTAL")));
    out->set_end_time (out->get_start_time () + f.number ("INTEGRATION_LENGTH"));
    unsigned* hits = out->get_hits ();
    for (unsigned ibin = 0; ibin < f.nbin; ibin++) hits[ibin] = f.hits[ibin];
    for (unsigned ichan = 0; ichan < f.nchan; ichan++)
      for (unsigned ipol = 0; ipol < f.npol; ipol++)
        memcpy (out->get_datptr (ichan, ipol), &f.sums[(size_t (ichan) * f.npol + ipol) * f.nbin * f.ndim],
                sizeof (float) * f.nbin * f.ndim);
  }
#endif
}

#endif
