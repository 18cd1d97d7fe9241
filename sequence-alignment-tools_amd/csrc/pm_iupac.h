// pm_iupac.h -- which stream characters a pattern character accepts under -w / -W.
// Data restated from the reference's table (util.cc:121-162): for every IUPAC letter the set of
// letters it is "compatible" with (its subsets and supersets, with the reference's own omissions:
// e.g. N lists V twice and not B).  A pattern character with no entry is matched literally.
#pragma once

namespace pm {

inline const char *iupac_compatible_set(unsigned char w) {
  switch (w) {
    case 'A': return "ARMWDHVN";      case 'B': return "GTUCYKSBN";
    case 'C': return "CYMSBHVN";      case 'D': return "GATURWKDN";
    case 'G': return "GRKSBDVN";      case 'H': return "ACTUMYWHN";
    case 'K': return "GTKBDN";        case 'M': return "ACMHVN";
    case 'N': return "ACGTURYKMSWVDHVN";
    case 'R': return "GARDVN";        case 'S': return "GCSBVN";
    case 'T': return "TUYKWVDHN";     case 'U': return "UTYKWVDHN";
    case 'V': return "GCARSMVN";      case 'W': return "ATUWDHN";
    case 'Y': return "TUCYBHN";       case 'X': return "MRWSYKVHDBXN";
    case 'a': return "armwdhvn";      case 'b': return "gtucyksbn";
    case 'c': return "cymsbhvn";      case 'd': return "gaturwkdn";
    case 'g': return "grksbdvn";      case 'h': return "actumywhn";
    case 'k': return "gtkbdn";        case 'm': return "acmhvn";
    case 'n': return "acgturykmswvdhvn";
    case 'r': return "gardvn";        case 's': return "gcsbvn";
    case 't': return "tuykwvdhn";     case 'u': return "utykwvdhn";
    case 'v': return "gcarsmvn";      case 'w': return "atuwdhn";
    case 'y': return "tucybhn";       case 'x': return "mrwsykvhdbxn";
  }
  return nullptr;
}

}  // namespace pm
